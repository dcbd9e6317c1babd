#!/bin/bash
# A/B runs on ONE box: launch length (pivots per resident-group launch) on the mid-size IP leg and the cold config-4 leg
mid() { python bench.py --only bnb_prune_mid --steps 1 --warmup 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); w=d['bnb_prune_mid']
print('mid', round(w['nodes_per_s']), w['lp_relaxations'], round(w['wall_s'],2))"; }
cold() { python tools/probe_regs.py 64 1600 2 | tail -1; }
for ch in 96 64 48 32 24; do echo "LPX_GROUP_CHUNK=$ch"; LPX_GROUP_CHUNK=$ch mid --bnb-mid-concurrent 256; LPX_GROUP_CHUNK=$ch cold; done
