#!/usr/bin/env python3
"""Per-category VALU instruction table of k_ntt_pass for one column of the bench step.

Compiles halo2_vectordb_amd/csrc/ntt.hip with --save-temps, splits the two step kernels (k_ntt_pass<LAST=false> / <LAST=true>) into
basic blocks, classifies every vector instruction of every block (inside an inline-asm product core: `core`; outside: add/sub,
shift/mask, multiply, move, compare/select), recognises the phases of a pass by their signature (number of product cores, LDS
traffic), and weights them with the number of times a wavefront executes them for ONE column of the C4 step —
lagrange_to_coeff at 2^16 (two 256-point passes) + coeff_to_extended to 2^18 (two 512-point passes, the first zero padded) —
following the host's schedule in ntt_dev (carry passes, Shoup range, the product-free block 0).  The total is held against the
measured SQ_INSTS_VALU per field product (profiles/r03g_sq_valu_summary.csv: 264.1).

usage: python tools/ntt_isa_table.py [--md profiles/r04_ntt_isa_table.md]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2_vectordb_amd", "csrc")

ADD = {"v_add_u32_e32", "v_add3_u32", "v_sub_u32_e32", "v_add_u32_e64", "v_sub_u32_e64", "v_subrev_u32_e32", "v_add_co_u32_e32", "v_addc_co_u32_e32", "v_add_co_u32_e64",
       "v_addc_co_u32_e64", "v_sub_co_u32_e32", "v_subb_co_u32_e32", "v_sub_co_u32_e64", "v_subb_co_u32_e64", "v_lshl_add_u64", "v_lshl_add_u32", "v_subrev_co_u32_e32",
       "v_subbrev_co_u32_e32"}


def category(op):
    if op in ADD:
        return "addsub"
    if re.match(r"v_(and|lshr|lshl|bfe|alignbit|or|bitop|bfi|perm|xor|not|bfrev)", op):
        return "shiftmask"
    if re.match(r"v_(mad|mul)", op):
        return "mul"
    if re.match(r"v_(mov|accvgpr)", op):
        return "mov"
    if re.match(r"v_(cmp|cndmask)", op):
        return "cmpsel"
    return "other"


def blocks_of(asm_lines, needle):
    funcs, cur = {}, None
    for i, l in enumerate(asm_lines):
        m = re.match(r"^(_ZN3vdb\w+):", l)
        if m:
            cur = m.group(1)
            funcs[cur] = [i, None]
        if l.startswith(".Lfunc_end") and cur:
            funcs[cur][1] = i
            cur = None
    name = [f for f in funcs if needle in f][0]
    lo, hi = funcs[name]
    out, label, body = [], "entry", []
    for l in asm_lines[lo:hi]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            out.append((label, body))
            label, body = m.group(1), []
        else:
            body.append(l)
    out.append((label, body))
    res = []
    for label, body in out:
        c, in_asm = collections.Counter(), False
        for l in body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                c["asm"] += 1
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            op = t.split()[0]
            if in_asm:
                if op.startswith("v_"):
                    c["core"] += 1
            elif op.startswith("v_"):
                c[category(op)] += 1
                c["valu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                c["vmem"] += 1
        res.append((label, c))
    return res


def phases(blocks):
    """name -> Counter, recognised by signature in source order"""
    P = {}
    four = [c for _l, c in blocks if c["asm"] == 4]
    assert len(four) == 6, [dict(c) for c in four]
    for name, c in zip(("r4_shoup_carry", "r4_shoup", "r4_shoup_s2", "r4_mont_carry", "r4_mont", "r4_mont_s2"), four):
        P[name] = c
    idx = {id(c): i for i, (_l, c) in enumerate(blocks)}
    P["r4_shoup_s2_blk0"] = blocks[idx[id(P["r4_shoup_s2"])] + 1][1]
    P["r4_mont_s2_blk0"] = blocks[idx[id(P["r4_mont_s2"])] + 1][1]
    first = [c for _l, c in blocks if c["asm"] == 1 and c["lds"] >= 20]
    assert len(first) == 2
    P["first_carry"], P["first"] = first
    # radix-2 stage: carry (no core, ~48 shift/mask + add), product (core, 3 LDS reads), add/sub + 2 puts (6 LDS)
    r2 = [c for _l, c in blocks if c["asm"] == 1 and c["lds"] == 3 and c["valu"] < 20]
    assert len(r2) == 1
    P["r2_product"] = r2[0]
    P["r2_addsub"] = [c for _l, c in blocks if c["asm"] == 0 and c["lds"] == 6 and c["vmem"] == 0 and 20 < c["valu"] < 40][0]
    P["r2_carry"] = [c for _l, c in blocks if c["asm"] == 0 and c["lds"] == 0 and c["vmem"] == 0 and 45 <= c["valu"] <= 52 and c["addsub"] == 16][0]
    P["load_coset_product"] = [c for _l, c in blocks if c["asm"] == 1 and c["lds"] == 0 and c["vmem"] == 0 and c["mov"] >= 10][0]
    P["twiddle_split"] = [c for _l, c in blocks if c["asm"] == 0 and c["shiftmask"] == 28 and c["mov"] >= 14][0]
    return P


def main():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=off",
                               "-I", CSRC, "-I", os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(CSRC, "ntt.hip"), "-o", os.path.join(d, "ntt.o")],
                              cwd=d, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(d, "ntt-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    nonlast, last = blocks_of(asm, "k_ntt_passILb0ELb0E"), blocks_of(asm, "k_ntt_passILb1ELb0E")
    PN, PL = phases(nonlast), phases(last)
    # write-out per element: non-last = a Montgomery product by the inter-pass twiddle (split of the twiddle, pack of the product); last
    # pass of an inverse multi-pass transform and of a forward transform = the multiplication-free reduction l9_canon_wide
    wo_n = [c for _l, c in nonlast if c["asm"] == 1 and c["vmem"] == 2 and c["lds"] == 4]
    write_nonlast = collections.Counter()
    for c in wo_n:
        write_nonlast.update(c)
    for k in write_nonlast:
        write_nonlast[k] /= len(wo_n)
    canon = [c for _l, c in last if c["asm"] == 0 and c["valu"] > 100 and c["mul"] >= 9][0]      # l9_canon_wide
    store = [c for _l, c in last if c["asm"] == 0 and c["vmem"] == 1 and c["cmpsel"] >= 9 and c["shiftmask"] < 10][0]
    index = [c for _l, c in last if c["asm"] == 0 and c["vmem"] == 2 and c["lds"] == 4][0]
    write_last = collections.Counter()
    for c in (canon, store, index):
        write_last.update(c)
    CARRY = 24.0      # l9_renorm of one element (PN["r4_shoup_carry"] - PN["r4_shoup"] over four elements)
    # ---- executions per column, in wave-instructions per THREAD-iteration (every lane of a wave runs the same block)
    rows = []         # (phase, what, count per column, Counter per execution, products per execution)

    def add(what, count, c, products, extra_shiftmask=0.0):
        c = collections.Counter(c)
        c["shiftmask"] += extra_shiftmask
        c["valu"] += extra_shiftmask
        rows.append((what, count, c, products))

    def passes(n, S, nonl, s0, coset, shoup_max_s, carry_steps, carry_out, P):
        tiles_elems = n            # elements per column in this pass
        r4_iters = tiles_elems / 4
        load = collections.Counter(valu=25.0, shiftmask=18.0, addsub=5.0, other=2.0)     # fetch address + l9_split per loaded element
        add(f"tile load (split) S={S}", tiles_elems >> s0, load, 0)
        if coset:
            add(f"coset factor on load S={S}", (tiles_elems >> s0) * 2 / 3, P["load_coset_product"], 1)
        st, step = s0, 0
        while st < S:
            carry = step in carry_steps
            if st + 1 < S:
                if st == 0:
                    add(f"radix-4 first step{' + carry' if carry else ''} S={S}", r4_iters, P["first_carry" if carry else "first"], 1)
                else:
                    kind = "shoup" if st <= shoup_max_s else "mont"
                    if st == 2 and not carry:        # block 0 takes the product-light body: a quarter of the wavefronts
                        add(f"radix-4 s=2 general ({kind}) S={S}", r4_iters * 3 / 4, P[f"r4_{kind}_s2"], 4)
                        add(f"radix-4 s=2 block 0 ({kind}) S={S}", r4_iters / 4, P[f"r4_{kind}_s2_blk0"], 1)
                    else:
                        add(f"radix-4 s={st} ({kind}){' + carry' if carry else ''} S={S}", r4_iters, P[f"r4_{kind}_carry" if carry else f"r4_{kind}"], 4)
                st += 2
            else:
                it = tiles_elems / 2
                if carry:
                    add(f"radix-2 carry S={S}", it, P["r2_carry"], 0)
                add(f"radix-2 product S={S}", it, P["r2_product"], 1)
                add(f"radix-2 add/sub S={S}", it, P["r2_addsub"], 0)
                st += 1
            step += 1
        if nonl:
            add(f"write-out: inter-pass twiddle product S={S}", tiles_elems, write_nonlast, 1, CARRY if carry_out else 0.0)
        else:
            add(f"write-out: reduction without a product S={S}", tiles_elems, write_last, 0)
        # stage-twiddle tables into LDS: m / 2 Montgomery entries split, per tile of 1024 elements
        add(f"stage twiddles to LDS S={S}", tiles_elems / 1024 * (1 << (S - 1)), P["twiddle_split"], 0)

    n16, n18 = 1 << 16, 1 << 18
    passes(n16, 8, True, 0, False, 6, {2}, True, PN)        # lagrange_to_coeff pass 0
    passes(n16, 8, False, 0, False, 6, {2}, False, PL)      # pass 1
    passes(n18, 9, True, 2, True, 5, {2}, False, PN)        # coeff_to_extended pass 0 (steps s = 2, 4, 6, 8; carry at the third)
    passes(n18, 9, False, 0, False, 5, {2, 4}, False, PL)   # pass 1 (carry at s = 4 and at the radix-2 stage)
    tot = collections.Counter()
    products = 0.0
    for what, count, c, prod in rows:
        for k in ("core", "addsub", "shiftmask", "mul", "mov", "cmpsel", "other"):
            tot[k] += count * c[k]
        products += count * prod
    lines = []
    lines.append("| phase | thread-executions per column | core | add/sub | shift/mask | other VALU | products |")
    lines.append("|---|---|---|---|---|---|---|")
    for what, count, c, prod in rows:
        lines.append(f"| {what} | {count:,.0f} | {c['core']:.0f} | {c['addsub']:.0f} | {c['shiftmask']:.0f} | {c['mul'] + c['mov'] + c['cmpsel'] + c['other']:.0f} | {prod} |")
    total = sum(tot.values())
    lines.append("")
    lines.append(f"products per column (model): {products:,.0f};  VALU instructions per column (per lane): {total:,.0f};  per product: **{total / products:.1f}**")
    lines.append("")
    lines.append("| category | instructions per product | share |")
    lines.append("|---|---|---|")
    for k, label in (("core", "product cores (inline asm: Montgomery 206, Shoup 179)"), ("addsub", "add / subtract (butterflies with their 14 r offsets, carries' adds, addressing)"),
                     ("shiftmask", "shift / mask (carry passes, limb split / pack, LDS addressing)"), ("mul", "multiplies outside the cores (reduction q r, index arithmetic)"),
                     ("mov", "moves"), ("cmpsel", "compare / select (conditional subtractions, bounds)"), ("other", "other")):
        lines.append(f"| {label} | {tot[k] / products:.1f} | {100 * tot[k] / total:.1f} % |")
    # ---- the same instructions by what they are FOR (the floors' table of profiles/r04_ntt_isa_table.md)
    noncore = lambda c: c["valu"] if "valu" in c else sum(c[k] for k in ("addsub", "shiftmask", "mul", "mov", "cmpsel", "other"))
    purpose = collections.Counter()
    for what, count, c, prod in rows:
        nc = sum(c[k] for k in ("addsub", "shiftmask", "mul", "mov", "cmpsel", "other"))
        purpose["product cores"] += count * c["core"]
        if what.startswith("radix-4"):
            arith = 108.0                                         # 4 carry-free additions (9) + 4 subtractions with the offset (18)
            if "block 0" in what or "first step" in what:
                arith = 108.0 + 9.0 * (3 if "block 0" in what else 1)   # + the carries of the subtrahends that are sums (l9_carry: 9 each... counted with the butterfly)
            carry = 4 * CARRY if "+ carry" in what else 0.0
            purpose["butterfly add / subtract"] += count * arith
            purpose["carry passes"] += count * carry
            purpose["LDS addressing, loop control in the steps"] += count * max(nc - arith - carry, 0.0)
        elif what.startswith("radix-2 add/sub"):
            purpose["butterfly add / subtract"] += count * 27.0
            purpose["LDS addressing, loop control in the steps"] += count * max(nc - 27.0, 0.0)
        elif what.startswith("radix-2 carry"):
            purpose["carry passes"] += count * nc
        elif what.startswith("radix-2 product"):
            purpose["LDS addressing, loop control in the steps"] += count * nc
        elif what.startswith("tile load") or what.startswith("coset factor"):
            purpose["tile load: limb split, coset factor set-up"] += count * nc
        elif what.startswith("write-out: inter-pass"):
            purpose["write-out with a product (non-last passes): twiddle split, pack, addresses, carry"] += count * nc
        elif what.startswith("write-out: reduction"):
            purpose["write-out without a product (last passes): l9_canon_wide, scatter address"] += count * nc
        elif what.startswith("stage twiddles"):
            purpose["stage twiddles into LDS"] += count * nc
    lines.append("")
    lines.append("| what the instructions are for | per product | share |")
    lines.append("|---|---|---|")
    for k, v in purpose.items():
        lines.append(f"| {k} | {v / products:.1f} | {100 * v / total:.1f} % |")
    text = "\n".join(lines)
    print(text)
    if "--md" in sys.argv:
        path = sys.argv[sys.argv.index("--md") + 1]
        with open(path, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
