"""Static instruction breakdown of the NTT pass kernels from the gfx950 assembly (measurement tool, runs without a GPU):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S eigen_zeth_amd/csrc/ntt.hip -o /tmp/ntt.s
    python tools/isa_breakdown.py /tmp/ntt.s > profiles/r4_ntt_isa_breakdown.txt
For every instantiation of the default 2^24 plan: the main loop (the largest backward-branch region: one tile = 16 elements per lane) is cut
out and its instructions are binned; counts are per ELEMENT per pass (loop body / 16), the unit of profiles/r3_integer_roofline.json
(PMC SQ_INSTS_VALU x 64 / elements).  Classes follow tools/ubench_isa.hip: everything in the VALU column issues at 4 cycles per wave
instruction except the plain 32-bit add / mov / logic forms (2 cycles)."""
import collections, re, sys

SHAPES = {"first pass (transposing, full twiddle table)": "ILi4ELi4ELi0ELi4ELb1ELb0ELi3ELb0E",
          "middle pass (per-tile twiddle table)": "ILi4ELi4ELi0ELi4ELb0ELb0ELi1ELb0E",
          "last pass (plain)": "ILi4ELi4ELi0ELi4ELb0ELb0ELi0ELb0E",
          "LDE: zero-padded first pass of the forward transform (2^25: radix 512)": "ILi4ELi3ELi0ELi5ELb1ELb1ELi3ELb0E",
          "LDE: last pass of the inverse transform (coset powers folded in)": "ILi4ELi4ELi0ELi4ELb0ELb0ELi2ELb0E"}
# every shape in both arithmetic forms: limbs of gl_limb.hpp (the default, knob ntt_limb) and the canonical carry chains of gl_asm.hpp
KERNELS = {}
for _t, _k in SHAPES.items():
    KERNELS[_t + " -- limb form"] = _k + "Lb1EE"
    KERNELS[_t + " -- canonical form"] = _k + "Lb0EE"


def classify(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "valu: v_mad_u64_u32 / v_mad_i64_i32 (32x32+64 multiply-add)"
    if re.match(r"v_(add|sub|subrev)(c|b)?_co_", op) or re.match(r"v_(addc|subb|subbrev)_", op):
        return "valu: add / sub with carry (chains of the modular add, sub, reduction)"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"):
        return "valu: compare / select"
    if re.match(r"v_(lshl|lshr|ashr|alignbit|lshlrev|lshrrev|lshl_add|lshl_or|and_or|bfe|bfi|perm)", op):
        return "valu: shifts / funnel shifts (the x 2^(12e) butterflies' twiddles, address bits)"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "valu: lane <-> scalar moves"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "valu: moves"
    if op.startswith("v_"):
        return "valu: other 32-bit (add, and, or, xor, mul_lo ...)"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "global memory"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "salu: waits / nops / barriers"
    if op.startswith("s_"):
        return "salu: other (addresses, loop, carries held in SGPR pairs)"
    return "other"


def main(path):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_Z") and l.rstrip().endswith(")") is False and ":" in l and "ntt_pass2_kernel" in l.split(":")[0]]
    for title, key in KERNELS.items():
        cand = [(i, n) for i, n in starts if key in n]
        if not cand:
            print("== %s: instantiation %s not in the assembly" % (title, key))
            continue
        i0, name = cand[0]
        i1 = next(j for j in range(i0 + 1, len(lines)) if lines[j].startswith("\t.section") or lines[j].startswith(".Lfunc_end"))
        body = lines[i0:i1]
        labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        best = None
        for k, l in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                n = sum(1 for x in body[labels[m.group(1)]:k + 1] if re.match(r"^\s+[a-z]", x) and not x.strip().startswith(";"))
                if best is None or n > best[0]:
                    best = (n, labels[m.group(1)], k)
        n, a, b = best
        ops = [x.split()[0] for x in body[a:b + 1] if re.match(r"^\s+[a-z]", x)]
        bins = collections.Counter(classify(o) for o in ops)
        valu = sum(v for k, v in bins.items() if k.startswith("valu"))
        vg = next((l for l in lines[i1:i1 + 400] if "NumVgprs" in l or ".vgpr_count" in l), "")
        print("== %s\n   %s\n   main loop: %d instructions per tile of 16 elements per lane; VALU %d = %.1f per element  %s" % (title, name[:120], len(ops), valu, valu / 16.0, vg.strip()))
        for k, v in sorted(bins.items(), key=lambda kv: -kv[1]):
            print("     %-90s %5d  %6.2f / element" % (k, v, v / 16.0))
        top = collections.Counter(o for o in ops if o.startswith("v_")).most_common(12)
        print("     most frequent VALU mnemonics: " + ", ".join("%s x%d" % kv for kv in top))


if __name__ == "__main__":
    main(sys.argv[1])
