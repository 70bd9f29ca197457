"""VALU instructions per unit of work for the integer-bound stages, from rocprofv3 --pmc SQ_INSTS_VALU summaries
(tools/pmc_summary.py output) -> profiles/<tag>_integer_roofline.json, which bench.py reads for the `roofline.integer` block
of the bench line ("report the integer-VALU fraction next to the HBM one").  Measurement tool.

usage: python tools/integer_roofline.py OUT.json ntt=SQ_NTT.json ntt_elems_log=28 poseidon=SQ_POSEIDON.json [stark=SQ_STARK.json] [msm=SQ_MSM.json]
Units: NTT pass = field elements of ONE LAUNCH = 2^ntt_elems_log, which the caller states (tools/pmc_ntt.py: 2^24 rows x 16 columns = 2^28 since
the launches were widened in round 4; rounds 1-3: 8 columns = 2^27.  Round 4's file divided by the old 2^27 and printed twice the true
count: the tool now refuses to guess, and cross-checks the figure against the launch's wave count); Poseidon = permutations
(hash_bench.py: 2^22 x 32 commit and a 2^22-state batch); quotient = LDE rows (stark_bench.py chunk64 2^20: 2^21 rows);
MSM bucket kernel = point additions of the launch are not counted by the profiler, so the MSM row is per input point."""
import json
import sys

out_path = sys.argv[1]
src = dict(a.split("=", 1) for a in sys.argv[2:])
ntt_elems_log = int(src.pop("ntt_elems_log")) if "ntt" in src else None      # KeyError: say how many elements a launch of the PMC run had
res = {"cycles_per_valu_instruction": 4.0,
       "note": "SQ_INSTS_VALU counts wave instructions: x 64 lanes / units of the launch = lane instructions per unit; every hot "
               "instruction of these kernels issues in 4 cycles per wave per SIMD (tools/ubench_isa.hip)",
       "sources": src, "ntt_valu_per_element_per_pass": {}, "stages": {}}


def valu(d, name):
    return d[name]["SQ_INSTS_VALU"]["mean"], d[name]["SQ_INSTS_VALU"]["mean_ms_under_profiler"]


if "ntt" in src:
    d = json.load(open(src["ntt"]))
    for k in d:
        if k.startswith("ntt_pass2_kernel"):
            v, ms = valu(d, k)
            # every lane of a pass kernel carries 16 elements per tile, a workgroup walks 1, 2 or 4 tiles: waves x 64 x 16 x tiles = elements
            waves = d[k].get("SQ_WAVES", {}).get("mean")
            if waves:
                tiles = (1 << ntt_elems_log) / (waves * 64 * 16)
                assert tiles in (1.0, 2.0, 4.0, 8.0), "ntt_elems_log=%d does not match the launch (%g waves): %g tiles per workgroup" % (ntt_elems_log, waves, tiles)
            res["ntt_valu_per_element_per_pass"][k] = v * 64 / float(1 << ntt_elems_log)
    res["ntt_elements_per_launch_log2"] = ntt_elems_log
if "poseidon" in src:
    d = json.load(open(src["poseidon"]))
    for k, units, label in (("poseidon_perm_kernel<true>", 1 << 22, "poseidon_perm (batch of 2^22 states)"),
                            ("merkle_leaves_kernel<true>", 4 * (1 << 22), "merkle leaves 2^22 x 32 (4 permutations per leaf)")):
        if k in d:
            v, ms = valu(d, k)
            res["stages"][label] = {"kernel": k, "valu_per_unit": v * 64 / units, "unit": "permutation", "ms_under_profiler": ms}
    lv = [k for k in d if k.startswith("merkle_level_kernel")]
    if lv:
        # launches cover levels of 2^21, 2^20, ... nodes: the mean launch has (sum of level sizes) / launches permutations
        n = d[lv[0]]["SQ_INSTS_VALU"]["launches"]
        v, ms = valu(d, lv[0])
        res["stages"]["merkle tree levels"] = {"kernel": lv[0], "valu_per_wave_instruction_mean_per_launch": v, "launches": n}
if "stark" in src:
    d = json.load(open(src["stark"]))
    for k in d:
        if "quotient" in k and "SQ_INSTS_VALU" in d[k]:
            v, ms = valu(d, k)
            res["stages"]["constraint quotient: " + k] = {"kernel": k, "valu_per_unit": v * 64 / float(d[k]["grid"]), "unit": "LDE row (grid size = rows)",
                                                          "ms_under_profiler": ms}
if "msm" in src:
    d = json.load(open(src["msm"]))
    tot = sum(d[k]["SQ_INSTS_VALU"]["mean"] * d[k]["SQ_INSTS_VALU"]["launches"] for k in d if "SQ_INSTS_VALU" in d[k])
    runs = 3.0       # msm_bench.py: one warm-up + two timed MSMs of 2^22 points
    res["stages"]["BN254 MSM G1 2^22 points (all kernels of one MSM)"] = {"valu_per_unit": tot * 64 / runs / float(1 << 22), "unit": "input point",
                                                                         "kernels": {k: d[k]["SQ_INSTS_VALU"]["mean"] for k in d if "SQ_INSTS_VALU" in d[k]}}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["ntt_valu_per_element_per_pass"], indent=1))
