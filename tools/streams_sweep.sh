#!/bin/bash
# 16-chunk batch (pre-generated witnesses) against the number of proving streams (measurement tool)
for s in 2 4 6 8 12; do
  echo "== prover_streams $s"
  ZP_PREGEN=1 ZP_STREAMS=$s python tools/batch_bench.py 16 20 chunk64 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('rep', d['rep'], 'batch_wall_s', round(d['batch_wall_s'], 3), 'chunk_proofs_s', round(d['chunk_proofs_s'], 3), 'final_s', round(d['aggregate_final_s'], 3))"
done
