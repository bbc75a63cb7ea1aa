#!/usr/bin/env python3
"""Assembly peephole for gfx950 (round 4 experiment): second and later reads of one VCC value by v_cndmask_b32.

Measured (tools/valu_rate_bench.hip, profiles/r04_valu_rates.txt): a v_cndmask_b32 that reads VCC is fast only as the FIRST
reader after the v_cmp that wrote it; every further v_cndmask on the same VCC value takes ~23 cycles (one compare + four
selects: 31 ns per group and SIMD against 4.8 ns for one compare + two selects on an SGPR pair).  Every 64-bit select -- a
double chosen by a predicate -- is such a pair of reads.  The same select through an explicit SGPR pair (VOP3 encoding) runs
at the normal rate.

The compiler keeps s[100:101] free of live values in this module (it only appears in the three-instruction
`s_or_saveexec_b64 s[100:101], -1 ... s_mov_b64 exec, s[100:101]` spill sequences).  This script copies VCC there right after a
compare whose value at least two v_cndmask_b32 read inside the same basic block, and points the second and later readers at the
copy:

        v_cmp_lt_f64_e32 vcc, v[2:3], v[4:5]             v_cmp_lt_f64_e32 vcc, v[2:3], v[4:5]
        v_cndmask_b32_e32 v0, v6, v8, vcc          ->      s_mov_b64 s[100:101], vcc
        v_cndmask_b32_e32 v1, v7, v9, vcc                  v_cndmask_b32_e32 v0, v6, v8, vcc
                                                           v_cndmask_b32_e64 v1, v7, v9, s[100:101]

VCC itself is left as it was (later readers of other kinds, branches, live-out values see what they saw).  Readers whose first
source is an SGPR or a literal keep VCC (the VOP3 form may read one scalar operand only on gfx9, and no literal).

usage: vcc_peephole.py in.s out.s      (prints what it rewrote)
"""
import re
import sys

SCR = "s[100:101]"
INLINE = re.compile(r"^(-?\d+|0x[0-9a-fA-F]+|-?\d+\.\d+|v\d+|v\[\d+:\d+\]|\|v\d+\||-v\d+)$")


def inline_const(tok):
    """True for VGPRs and inline constants (integers -16..64, a handful of floats); False for SGPRs and literals."""
    tok = tok.strip()
    if re.match(r"^v\d+$", tok):
        return True
    if re.match(r"^-?\d+$", tok):
        return -16 <= int(tok) <= 64
    return tok in ("0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0")


def writes_vcc(op, ops):
    first = ops.split(",")[0].strip() if ops else ""
    if first in ("vcc", "vcc_lo", "vcc_hi"):
        return True
    # carry-out / implicit writers: v_add_co_u32 vD, vcc, ...   v_div_scale_* vD, vcc, ...   v_readlane etc. never write vcc
    parts = [p.strip() for p in ops.split(",")]
    if op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")):
        return len(parts) > 1 and parts[1] == "vcc"
    return False


def main():
    src, dst = sys.argv[1], sys.argv[2]
    only = re.compile(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None   # functions to rewrite (regex on the mangled name)
    skip = re.compile(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else None
    lines = open(src).read().split("\n")
    fn_of = []
    cur = ""
    for ln in lines:
        mf = re.match(r"^(_Z[\w$.]+):", ln)
        if mf:
            cur = mf.group(1)
        fn_of.append(cur)
    out = []
    n = len(lines)
    i = 0
    groups = converted = 0
    while i < n:
        ln = lines[i]
        t = ln.strip()
        m = re.match(r"^(v_cmp\w*_e32)\s+vcc,", t)
        if m and ((only and not only.search(fn_of[i])) or (skip and skip.search(fn_of[i]))):
            m = None
        if not m or "sdwa" in t:
            out.append(ln)
            i += 1
            continue
        # scan the rest of the basic block for readers of this vcc value
        j = i + 1
        readers = []
        while j < n:
            u = lines[j].strip()
            if not u or u.startswith(";") or u.startswith("."):
                if u.startswith(".LBB") or re.match(r"^[.\w$]+:", u):
                    break  # a label: block ends
                j += 1
                continue
            if re.match(r"^[.\w$]+:", u):
                break
            op = u.split()[0]
            ops = u[len(op):].split(";")[0].strip()
            if "s[100:101]" in u or re.search(r"\bs10[01]\b", u):
                break  # the compiler's own use of the pair: stop here
            if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier", "s_call")):
                break
            if writes_vcc(op, ops):
                # an instruction may read the old value and write the new one (v_addc_co): it is not a cndmask, stop either way
                break
            mm = re.match(r"^v_cndmask_b32_e(32|64)\s+(v\d+),\s*([^,]+),\s*([^,]+),\s*vcc$", u.split(";")[0].strip())
            if mm:
                readers.append((j, mm))
            j += 1
        conv = []
        for k, (idx, mm) in enumerate(readers):
            if k == 0:
                continue
            s0, s1 = mm.group(3).strip(), mm.group(4).strip()
            if inline_const(s0) and inline_const(s1):
                conv.append((idx, mm))
        out.append(ln)
        if conv:
            indent = ln[: len(ln) - len(ln.lstrip())]
            out.append(f"{indent}s_mov_b64 {SCR}, vcc")
            groups += 1
            cset = {idx: mm for idx, mm in conv}
            for q in range(i + 1, j):
                if q in cset:
                    mm = cset[q]
                    l2 = lines[q]
                    ind2 = l2[: len(l2) - len(l2.lstrip())]
                    out.append(f"{ind2}v_cndmask_b32_e64 {mm.group(2)}, {mm.group(3).strip()}, {mm.group(4).strip()}, {SCR}")
                    converted += 1
                else:
                    out.append(lines[q])
            i = j
        else:
            i += 1
    open(dst, "w").write("\n".join(out))
    print(f"vcc_peephole: {groups} compares copied to {SCR}, {converted} v_cndmask_b32 readers moved off VCC")


if __name__ == "__main__":
    main()
