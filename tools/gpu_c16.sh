#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "gemm_nt" 2>&1 | tail -n 2
W="10688,3072,768"; Q="10688,2304,768"; A="10688,768,768"; B="10688,768,2304"; C="10688,768,3072"
bash tools/gpu_ab_env.sh "table (27/29/10):" "n768=59:VLP_NT_OVERRIDE=$A=59;$B=59;$C=59" "W=61:VLP_NT_OVERRIDE=$W=61" "n768=59 W=61:VLP_NT_OVERRIDE=$A=59;$B=59;$C=59;$W=61" "n768=59 W=61 Q=61:VLP_NT_OVERRIDE=$A=59;$B=59;$C=59;$W=61;$Q=61" "table again:"
