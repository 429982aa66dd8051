#!/bin/bash
export PYTHONUNBUFFERED=1
for st in 0 4000 8000 12000 16000 24000; do echo "stagger $st"; VLP_ATTN_STAGGER=$st python tools/attn_lab.py 2>/dev/null | grep "B= 64"; done
