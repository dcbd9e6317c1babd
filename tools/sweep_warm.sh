#!/bin/bash
# A/B runs of the warm B&B leg on ONE box (boxes differ by +-5 %): prints nodes/s and the whole-leg fraction per setting
run() { python bench.py --only bnb_warm --steps 1 --warmup 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); w=d['bnb_warm']
print(round(w['nodes_per_s']), round(w['roofline']['frac'],3))"; }
for i in 1 2; do
echo "default"; run
echo "sets 3"; LPX_ROLL_SETS=3 run
echo "sets 3 one stream"; LPX_ROLL_SETS=3 LPX_ROLL_ONE_STREAM=1 run
echo "sets 4 one stream"; LPX_ROLL_SETS=4 LPX_ROLL_ONE_STREAM=1 run
echo "sets 3 one stream conc 48"; LPX_ROLL_SETS=3 LPX_ROLL_ONE_STREAM=1 run --bnb-warm-concurrent 48
echo "sets 4 one stream conc 32"; LPX_ROLL_SETS=4 LPX_ROLL_ONE_STREAM=1 run --bnb-warm-concurrent 32
echo "sets 3 conc 48"; LPX_ROLL_SETS=3 run --bnb-warm-concurrent 48
done
