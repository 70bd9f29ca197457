#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box: tools/profile_round.sh TAG   (outputs under gpurun_out/)
# kernel stats of the bench command, SQ / FETCH / WRITE counters of the NTT passes (separate --pmc passes), SQ counters
# of the Poseidon kernels.  The program itself follows `--` (no env/bash hop under the profiler).
set -e
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_bench -- python3 $ROOT/bench.py --no-cpu --no-pipeline --steps 10 > $OUT/${TAG}_bench_under_profiler.json 2> $OUT/${TAG}_bench_under_profiler.err
cp $(ls $OUT/prof_${TAG}_bench/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_${TAG}_sq -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_sq ntt_pass2 > $OUT/${TAG}_pmc_sq_ntt.json
echo "sq ntt done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_fetch ntt_pass2 > $OUT/${TAG}_pmc_fetch_ntt.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_write ntt_pass2 > $OUT/${TAG}_pmc_write_ntt.json
echo "traffic done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_${TAG}_pos -- python3 $ROOT/tools/hash_bench.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_pos > $OUT/${TAG}_pmc_sq_poseidon.json
echo "poseidon done"
rm -rf $OUT/prof_${TAG}_*
