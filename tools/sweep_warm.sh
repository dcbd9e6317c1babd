#!/bin/bash
# A/B runs of the warm B&B leg on ONE box (boxes differ by +-5 %): prints nodes/s and the whole-leg fraction per setting
run() { python bench.py --only bnb_warm --steps 1 --warmup 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); w=d['bnb_warm']
print(round(w['nodes_per_s']), round(w['roofline']['frac'],3))"; }
for i in 1 2 3; do
echo "default"; run
echo "LPX_HANDLE_CACHE=0"; LPX_HANDLE_CACHE=0 run
echo "LPX_ROLL_ONE_STREAM=1"; LPX_ROLL_ONE_STREAM=1 run
done
