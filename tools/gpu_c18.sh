#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py tests/test_20_fullsize_gpu.py tests/test_40_decode_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 3
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items(): print(k, {a:round(b,5) for a,b in v.items() if 'fp16' in a})
PY
bash tools/gpu_ab_env.sh "skip on:" "skip off:VLP_ATTN_SKIP=0" "skip on again:" "skip off again:VLP_ATTN_SKIP=0"
echo "--- CC mixed masks"; for f in 1 0; do VLP_ATTN_SKIP=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --s2s_prob 0.75 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('skip=$f  %.3f ms/step' % d['ms_per_step'])"; done
echo "--- VQA bi masks"; for f in 1 0; do VLP_ATTN_SKIP=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tasks vqa2 --s2s_prob 0 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('skip=$f  %.3f ms/step' % d['ms_per_step'])"; done
