#!/bin/bash
# ONE parameterised GPU-box script (replaces the per-round gpu_r4_*.sh / gpu_r5_*.sh / gpu_final.sh / gpu_all.sh / gpu_bench_prof.sh /
# gpu_wp_probe*.sh / gpu_*_lab.sh one-offs).  Runs ON the GPU box:  gpurun --timeout N -- 'bash tools/gpu_run.sh <task> [args] [-- <task> ...]'
# Everything it writes goes to gpurun_out/ (merged back by gpurun); copy what should be judged into profiles/rNN_*.
#
#   suite [pytest args]        python -m pytest tests -m gpu -> gpurun_out/pytest_gpu_full.log (+ box header)
#   labsuite                   the investigation variants against vlp_amd/libvlp_hip_lab.so (python -m vlp_amd.build --lab first)
#   smoke                      __graft_entry__.smoke()
#   bench [TAG] [bench args]   python bench.py ... -> gpurun_out/bench_TAG.json
#   shapes                     CC mixed-mask and VQA bench lines (BASELINE configs[3], [4])
#   prof dense|varlen [steps]  rocprofv3 --kernel-trace --stats of the bench command (side stream off) + steady-state summary
#   pmc                        FETCH_SIZE / WRITE_SIZE / SQ counter passes of the bench command (tools/gpu_pmc_bench.sh)
#   ab "name:ENV=.." ...       same-box in-step A/B of environment switches (tools/gpu_ab_env.sh)
#   ntlab [variants] [rotate]  cold-operand NT GEMM lab        tnlab / attnlab / lnlab / adamlab / varlenlab: the other per-kernel labs
#   decode [MODES]             decoder bench + rocprofv3 steady-state summary of the greedy run
#   loader                     loader-only / resident / prefetcher-fed training rates (tools/loader_bench.py)
#   soak                       3 000-step sustained run between two 20-step runs (tools/gpu_soak.sh)
#   evidence                   suite + smoke + bench + shapes + prof dense + prof varlen + decode + loader  (round-end record on one box)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out
hdr() { echo "# box: $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Model' | sed 's/.*: *//') $(nproc) host threads, $(date -u +%FT%TZ), git $(cat .git_head 2>/dev/null)"; }

task_suite() { hdr > $OUT/pytest_gpu_full.log; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 "$@" >> $OUT/pytest_gpu_full.log 2>&1; echo "pytest exit $?"; tail -n 4 $OUT/pytest_gpu_full.log; }
task_labsuite() {
  [ -f vlp_amd/libvlp_hip_lab.so ] || { echo "no vlp_amd/libvlp_hip_lab.so (python -m vlp_amd.build --lab)"; return; }
  VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_lab.so timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "phased or stream_k or attention_fwd_bwd or epilogues or asymmetric" > $OUT/pytest_gpu_lab_variants.log 2>&1
  echo "lab pytest exit $?"; tail -n 2 $OUT/pytest_gpu_lab_variants.log; }
task_smoke() { timeout 300 python __graft_entry__.py > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 $OUT/smoke.log; }
task_bench() { local tag=${1:-default}; shift; timeout 600 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "bench exit $?"; cut -c1-2500 $OUT/bench_$tag.json; }
task_shapes() {
  timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-parity --s2s_prob 0.75 > $OUT/bench_cc.json 2>/dev/null; cut -c1-200 $OUT/bench_cc.json
  timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-parity --tasks vqa2 --s2s_prob 0 > $OUT/bench_vqa.json 2>/dev/null; cut -c1-200 $OUT/bench_vqa.json; }
task_prof() {
  local mode=${1:-dense} steps=${2:-20} v=0; [ "$mode" = varlen ] && v=1
  rm -rf /tmp/prof_$mode
  VLP_VARLEN=$v VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o p -- python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-parity --no-varlen > $OUT/prof_bench_line_$mode.json 2> $OUT/prof_$mode.err; echo "rocprof $mode exit $?"
  python tools/prof_summary.py $(find /tmp/prof_$mode -name "*kernel_trace.csv" | head -1) 0.45 > $OUT/prof_summary_$mode.txt 2>&1; head -n 24 $OUT/prof_summary_$mode.txt | cut -c1-170
  cp $(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1) $OUT/prof_kernel_stats_$mode.csv 2>/dev/null; }
task_pmc() { bash tools/gpu_pmc_bench.sh > $OUT/pmc_run.log 2>&1; tail -n 30 $OUT/pmc_run.log; }
task_ab() { bash tools/gpu_ab_env.sh "$@" 2>&1 | tee $OUT/ab.txt; }
task_ntlab() { timeout 600 python tools/nt_lab.py --rotate=${2:-12} --variants=${1:-27,29,73,77,264} 2>&1 | grep -v amdgpu.ids | tee $OUT/nt_lab.txt; }
task_tnlab() { timeout 300 python tools/tn_group_lab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/tn_lab.txt; }
task_attnlab() { timeout 300 python tools/attn_lab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_lab.txt; }
task_lnlab() { timeout 300 python tools/ln_lab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ln_lab.txt; }
task_adamlab() { timeout 300 python tools/adam_lab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/adam_lab.txt; }
task_varlenlab() { timeout 900 python tools/varlen_lab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/varlen_lab.txt; }
task_decode() {
  MODES=${1:-greedy,beam3} timeout 600 python tools/decode_bench.py > $OUT/decode_bench.json 2> $OUT/decode_bench.err; cat $OUT/decode_bench.json
  rm -rf /tmp/profd; MODES=greedy timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd -o dec -- python tools/decode_bench.py > $OUT/prof_decode.log 2> $OUT/prof_decode.err
  python tools/prof_summary.py /tmp/profd/dec_kernel_trace.csv 0.25 > $OUT/prof_decode_steady.txt; head -n 30 $OUT/prof_decode_steady.txt | cut -c1-170; }
task_loader() { N_IMAGES=${1:-2048} timeout 900 python tools/loader_bench.py > $OUT/loader_bench.json 2> $OUT/loader_bench.err; cat $OUT/loader_bench.json; }
task_soak() { bash tools/gpu_soak.sh > /dev/null 2>&1; grep -v "^    " $OUT/soak.txt; }
task_evidence() { task_suite; task_smoke; task_bench default; task_shapes; task_prof dense; task_prof varlen; task_decode; task_loader; }

[ $# -eq 0 ] && { sed -n 2,22p "$0"; exit 0; }
args=()
run() { [ ${#args[@]} -gt 0 ] && { local t=${args[0]}; echo "=== $t ${args[*]:1}"; "task_$t" "${args[@]:1}"; }; args=(); }
for a in "$@"; do if [ "$a" = "--" ]; then run; else args+=("$a"); fi; done
run
