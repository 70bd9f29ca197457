#!/bin/bash
# rocprofv3 per-kernel stats of (1) three chunk proofs (chunk64 x 2^20, the service's parameters) and (2) the recursion layers
# (GenAggregatedProof + GenFinalProof over two such proofs): tools/profile_proofs.sh TAG   (outputs under gpurun_out/)
set -e
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_chunk -- python3 $ROOT/tools/stark_bench.py chunk64 20 3 > $OUT/${TAG}_stark_bench.txt 2>&1
cp $(ls $OUT/prof_${TAG}_chunk/*/*kernel_stats.csv | head -1) $OUT/${TAG}_stark_chunk64_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_rec -- python3 $ROOT/tools/recursion_bench.py 20 3 > $OUT/${TAG}_recursion_under_profiler.txt 2>&1
cp $(ls $OUT/prof_${TAG}_rec/*/*kernel_stats.csv | head -1) $OUT/${TAG}_recursion_kernel_stats.csv
rm -rf $OUT/prof_${TAG}_chunk $OUT/prof_${TAG}_rec
echo done
