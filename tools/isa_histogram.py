"""Per-loop opcode histogram of a kernel's gfx950 ISA, priced with the measured issue rates of
profiles/r03_valu_rate_table.txt (tools/microbench/ub_ops: ~2.6 cycles per wave-instruction per SIMD for full-rate VALU
ops, ~4.6 for the half-rate ones), so that "x % of what this instruction mix can issue" is checkable:

    python tools/isa_histogram.py rust-bio_amd/csrc/sw_fill_pk16_local.hip 'sw_fill_pk16_kernelILi10ELi16' > profiles/...

Compiles the unit with -save-temps into a scratch directory, finds the kernel whose mangled name contains the pattern,
lists every loop (backward branch) with its instruction count by class and the VALU issue cycles of one trip."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# half-rate VALU ops on gfx950 (profiles/r03_valu_rate_table.txt): everything VOP3-only, v_max/min_i32/u32, shifts left,
# compares, selects, DPP moves, packed 16-bit ops
HALF = re.compile(r"^(v_max_[iu]32|v_min_[iu]32|v_max_f32|v_lshlrev_b32|v_lshlrev_b64|v_lshrrev_b64|v_cmp|v_cndmask|v_max3|v_min3|v_med3|v_bfi|v_and_or|v_or3|"
                  r"v_lshl_or|v_lshl_add|v_add3|v_add_lshl|v_xad|v_perm|v_mad_|v_mul_|v_bfe|v_alignb|v_sad|v_pk_|v_mov_b32_dpp|v_bitop3|"
                  r"v_mbcnt|v_readlane|v_writelane|v_bcnt|v_ffb|v_mbcnt)")
FULL_C, HALF_C = 2.65, 4.6


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "rust-bio_amd", "csrc"), "-c", os.path.abspath(src), "-save-temps",
                               "-o", os.path.join(d, "o.o")], cwd=d, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
        s = open(os.path.join(d, asm)).read()
    names = [n for n in re.findall(r"^(_Z\S+):", s, re.M) if pat in n]
    if not names:
        sys.exit("no kernel matches " + pat)
    name = names[0]
    i = s.index(name + ":")
    body = s[i:s.index(".Lfunc_end", i)].split("\n")
    meta = s[s.index(".name:           " + name):]
    vg = re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)
    sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)
    print(f"# {os.path.relpath(os.path.abspath(src), ROOT)} :: {name}")
    print(f"# vgpr_count {vg}, vgpr_spill_count {sp}; issue cycles per wave-instruction per SIMD: full-rate {FULL_C}, half-rate {HALF_C} "
          "(profiles/r03_valu_rate_table.txt)")
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < k:
                loops.append((labels[t], k, t))
    for a, b, t in sorted(set(loops)):
        ops = collections.Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ".;":
                continue
            op = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", l.split()[0])
            if "dpp" in l and op == "v_mov_b32":
                op = "v_mov_b32_dpp"
            ops[op] += 1
        valu = {o: c for o, c in ops.items() if o.startswith("v_")}
        half = sum(c for o, c in valu.items() if HALF.match(o))
        full = sum(valu.values()) - half
        other = collections.Counter()
        for o, c in ops.items():
            if not o.startswith("v_"):
                other[o.split("_")[0]] += c
        print(f"\nloop {t}: lines {a}..{b}, {sum(ops.values())} instructions: VALU {sum(valu.values())} (full-rate {full}, half-rate {half}) "
              f"= {full * FULL_C + half * HALF_C:.0f} issue cycles per trip; " + ", ".join(f"{k} {v}" for k, v in sorted(other.items())))
        print("   " + ", ".join(f"{o} {c}" for o, c in sorted(valu.items(), key=lambda x: -x[1])[:28]))


if __name__ == "__main__":
    main()
