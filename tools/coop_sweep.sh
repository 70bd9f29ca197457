#!/bin/bash
# 16-chunk batch against the size below which Merkle levels go to the 12-lanes-per-node kernels (measurement tool)
for c in 9 11 12 13 14 15; do
  echo "== merkle_coop_log $c"
  ZP_MERKLE_COOP_LOG=$c ZP_PREGEN=1 python tools/batch_bench.py 16 20 chunk64 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('rep', d['rep'], 'batch_wall_s', round(d['batch_wall_s'], 3), 'chunk_proofs_s', round(d['chunk_proofs_s'], 3))"
done
