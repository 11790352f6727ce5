#!/usr/bin/env python3
"""Issue-slot budget of a kernel's steady-state loop from a hipcc -S listing.

usage: tools/isa_budget.py file.s mangled_substring RANGES
  RANGES = comma-separated 'bin:first-last[:opcode-regex]' line ranges (1-based, relative to the kernel's first line) that
  attribute instructions of the per-row loop to a purpose; lines outside every range are ignored.
Every instruction is weighted with the SIMD cycles per wave-instruction measured on MI355X with
tools/ubench/isa_probe2/3 (4 wavefronts per SIMD, gpurun_out/isa_probe{2,3}.txt, summary in
profiles/r03_sad_isa_budget.md): 'fast' 32-bit VOP1/VOP2 adds/logic/moves ~2.9, every other VALU op ~4.7,
v_(m)qsad_pk_u16_u8 ~17.2; DS / VMEM / SALU instructions are listed by count (they issue beside the VALU)."""
import collections, re, sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_not_b32"}
QUARTER = {"v_mqsad_pk_u16_u8", "v_qsad_pk_u16_u8"}
COST = {"fast": 2.9, "slow": 4.7, "quarter": 17.2, "trans": 8.3}
TRANS = {"v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32"}

def klass(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_"):
        if base in QUARTER: return "quarter"
        if base in TRANS: return "trans"
        if base in FAST and not op.endswith(("_dpp", "_sdwa")): return "fast"
        return "slow"
    if base.startswith("ds_"): return "lds"
    if base.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if base.startswith("s_"):
        return "wait" if base in ("s_waitcnt", "s_nop", "s_barrier") else "salu"
    return "other"

def main():
    s = open(sys.argv[1]).read()
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(sys.argv[2]), s, re.M)
    body = s[m.end():s.index(".Lfunc_end", m.end())].splitlines()
    ranges = []
    for r in sys.argv[3].split(","):
        name, span, *flt = r.split(":")
        a, b = span.split("-")
        ranges.append((name, int(a), int(b), re.compile(flt[0]) if flt else None))
    bins = collections.OrderedDict()
    for name, a, b, flt in ranges:
        c = bins.setdefault(name, collections.Counter())
        for line in body[a - 1:b]:
            t = line.strip()
            if not t or t.startswith((".", ";")) or t.endswith(":"): continue
            if flt is not None and not flt.search(t.split()[0]): continue
            c[klass(t.split()[0])] += 1
    tot = collections.Counter()
    print("| bin | fast | slow | quarter | trans | VALU instr | VALU SIMD-cycles | DS | VMEM | SALU |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, c in bins.items():
        n = c["fast"] + c["slow"] + c["quarter"] + c["trans"]
        cyc = sum(c[k] * COST[k] for k in COST)
        tot.update(c); tot["cyc"] += cyc
        print(f"| {name} | {c['fast']} | {c['slow']} | {c['quarter']} | {c['trans']} | {n} | {cyc:.0f} | {c['lds']} | {c['vmem']} | {c['salu']} |")
    n = tot["fast"] + tot["slow"] + tot["quarter"] + tot["trans"]
    print(f"| **total** | {tot['fast']} | {tot['slow']} | {tot['quarter']} | {tot['trans']} | {n} | {tot['cyc']:.0f} | {tot['lds']} | {tot['vmem']} | {tot['salu']} |")

if __name__ == "__main__":
    main()
