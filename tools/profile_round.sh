#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box: tools/profile_round.sh TAG   (outputs under gpurun_out/)
# kernel stats of the bench command, SQ / FETCH / WRITE counters of the NTT passes (separate --pmc passes), SQ counters
# of the Poseidon kernels.  The program itself follows `--` (no env/bash hop under the profiler).
set -e
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_bench -- python3 $ROOT/bench.py --no-cpu --no-pipeline --no-config5 --steps 10 > $OUT/${TAG}_bench_under_profiler.json 2> $OUT/${TAG}_bench_under_profiler.err
cp $(ls $OUT/prof_${TAG}_bench/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_${TAG}_sq -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_sq ntt_pass2 > $OUT/${TAG}_pmc_sq_ntt.json
echo "sq ntt done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_fetch ntt_pass2 > $OUT/${TAG}_pmc_fetch_ntt.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_write ntt_pass2 > $OUT/${TAG}_pmc_write_ntt.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_lfetch -- python3 $ROOT/tools/pmc_lde.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_lfetch ntt_pass2 > $OUT/${TAG}_pmc_fetch_lde.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_lwrite -- python3 $ROOT/tools/pmc_lde.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_lwrite ntt_pass2 > $OUT/${TAG}_pmc_write_lde.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_lstats -- python3 $ROOT/tools/pmc_lde.py > /dev/null 2>&1
cp $(ls $OUT/prof_${TAG}_lstats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_lde_kernel_stats.csv
echo "traffic done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_${TAG}_pos -- python3 $ROOT/tools/hash_bench.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_pos > $OUT/${TAG}_pmc_sq_poseidon.json
echo "poseidon done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_${TAG}_stark -- python3 $ROOT/tools/stark_bench.py chunk64 20 1 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_stark quotient > $OUT/${TAG}_pmc_sq_quotient.json
echo "quotient done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_${TAG}_msm -- python3 $ROOT/tools/msm_bench.py 22 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_msm msm > $OUT/${TAG}_pmc_sq_msm.json
echo "msm done"
python3 $ROOT/tools/integer_roofline.py $OUT/${TAG}_integer_roofline.json ntt=$OUT/${TAG}_pmc_sq_ntt.json ntt_elems_log=28 poseidon=$OUT/${TAG}_pmc_sq_poseidon.json stark=$OUT/${TAG}_pmc_sq_quotient.json msm=$OUT/${TAG}_pmc_sq_msm.json > /dev/null
python3 - <<PY
import json
f = json.load(open("$OUT/${TAG}_pmc_fetch_ntt.json")); w = json.load(open("$OUT/${TAG}_pmc_write_ntt.json"))
rows = {k: {"fetch_bytes_x2": 2 * f[k]["FETCH_SIZE"]["mean"] * 1024, "write_bytes": w[k]["WRITE_SIZE"]["mean"] * 1024} for k in f if k in w}
worst = max(rows.values(), key=lambda r: r["fetch_bytes_x2"] + r["write_bytes"])
json.dump({"per_kernel": rows, "hbm_bytes_per_launch_dominant": worst["fetch_bytes_x2"] + worst["write_bytes"],
           "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (KiB units); FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md"},
          open("$OUT/${TAG}_ntt_traffic.json", "w"), indent=1)
PY
# the two-pass plan (two radix-4096 passes, 1024-thread workgroups, 128-KiB tiles of 4 columns: knob ntt_maxl = 12) under the same counters: what
# stops it -- instruction issue, LDS, or the bandwidth of its 32-byte runs (round-4 review item 3: the A/B with counters)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_${TAG}_sq2p -- python3 $ROOT/tools/pmc_ntt.py ntt_maxl=12 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_sq2p ntt_pass2 > $OUT/${TAG}_pmc_sq_ntt_two_pass.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/prof_${TAG}_lds2p -- python3 $ROOT/tools/pmc_ntt.py ntt_maxl=12 > /dev/null 2>&1 || true
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_lds2p ntt_pass2 > $OUT/${TAG}_pmc_lds_ntt_two_pass.json || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/prof_${TAG}_lds3p -- python3 $ROOT/tools/pmc_ntt.py > /dev/null 2>&1 || true
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_lds3p ntt_pass2 > $OUT/${TAG}_pmc_lds_ntt.json || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch2p -- python3 $ROOT/tools/pmc_ntt.py ntt_maxl=12 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_fetch2p ntt_pass2 > $OUT/${TAG}_pmc_fetch_ntt_two_pass.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write2p -- python3 $ROOT/tools/pmc_ntt.py ntt_maxl=12 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/prof_${TAG}_write2p ntt_pass2 > $OUT/${TAG}_pmc_write_ntt_two_pass.json
echo "two-pass counters done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_ood -- python3 $ROOT/tools/stark_bench.py chunk64 22 1 > $OUT/${TAG}_stark_chunk64_2p22.txt 2>&1
cp $(ls $OUT/prof_${TAG}_ood/*/*kernel_stats.csv | head -1) $OUT/${TAG}_stark_chunk64_2p22_kernel_stats.csv
echo "chunk proof stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_rec -- python3 $ROOT/tools/recursion_bench.py 20 3 > $OUT/${TAG}_recursion_under_profiler.txt 2>&1
cp $(ls $OUT/prof_${TAG}_rec/*/*kernel_stats.csv | head -1) $OUT/${TAG}_recursion_kernel_stats.csv
echo "recursion done"
rm -rf $OUT/prof_${TAG}_*
