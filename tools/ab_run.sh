#!/bin/bash
# A/B of library variants on the same box: tools/ab_run.sh <tool> <args...>  (variants: build_ab/*.so + the default)
set -e
for rep in 1 2; do
  for lib in "" $(ls build_ab/*.so 2>/dev/null); do
    echo "== rep $rep lib ${lib:-default}"
    if [ -z "$lib" ]; then python "$@"; else ZP_LIB_PATH=$PWD/$lib python "$@"; fi
  done
done
