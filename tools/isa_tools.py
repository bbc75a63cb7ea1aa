#!/usr/bin/env python3
"""Diagnostics on the device assembly of libmdt_hip.so (round 4, the "ISA pass"; no GPU needed).

    mkdir -p /tmp/isa && cd /tmp/isa && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -I $REPO/include \\
        -save-temps -o lib.so $REPO/mcmcdiagnostictools.jl_amd/csrc/mdt_hip.hip          # writes mdt_hip-hip-amdgcn-amd-amdhsa-gfx950.s
    python tools/isa_tools.py spills  FILE.s                 # scratch reloads INSIDE loops, per function (a reload behind an
                                                             #   s_waitcnt vmcnt(0) in a hot loop is a memory round trip per element)
    python tools/isa_tools.py func    FILE.s MANGLED_NAME    # one function's assembly to stdout
    python tools/isa_tools.py hist    FILE.s FIRST LAST      # instructions by class (VALU / SALU / LDS / VMEM) and VALU opcode in a line range
    python tools/isa_tools.py loops   FILE.s                 # outermost loops with more than 1500 instructions: line range, size
"""
import collections
import re
import sys


def spills(path):
    depth_re = re.compile(r'in Loop: Header=\S+ Depth=(\d+)|Loop Header: Depth=(\d+)')
    lines = open(path).read().split('\n')
    res, cur, last_depth = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            cur, last_depth = m.group(1), 0
            continue
        if l.startswith('.LBB') or '; %bb.' in l:
            d = 0
            for ll in lines[i:i + 3]:
                mm = depth_re.search(ll)
                if mm:
                    d = int(mm.group(1) or mm.group(2))
                    break
            last_depth = d
        if 'scratch_load' in l and cur and last_depth >= 1:
            res.setdefault(cur, collections.Counter())[last_depth] += 1
    for k, v in sorted(res.items(), key=lambda kv: -sum(kv[1].values())):
        print(sum(v.values()), dict(v), k[:120])


def func(path, name):
    on = False
    for l in open(path):
        if l.startswith(name + ':'):
            on = True
        if on:
            sys.stdout.write(l)
            if l.startswith('.Lfunc_end'):
                break


def hist(path, a, b):
    c = collections.Counter()
    for i, l in enumerate(open(path), 1):
        if i < a or i >= b:
            continue
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'):
            continue
        op = l.split()[0]
        k = 'VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'LDS' if op.startswith('ds_') else \
            'VMEM:' + op.split('_')[1] if op.split('_')[0] in ('global', 'scratch', 'buffer', 'flat') else op
        c[k] += 1
        if k == 'VALU':
            c['  ' + op] += 1
    for k, v in sorted(c.items(), key=lambda x: -x[1])[:40]:
        print(f"{v:6d} {k}")


def loops(path):
    lines = open(path).read().split('\n')
    first, last = collections.defaultdict(lambda: 10 ** 9), collections.defaultdict(int)
    for i, l in enumerate(lines):
        m = re.search(r'Header=(BB\d+_\d+) Depth=1', l)
        if m:
            h = m.group(1)
            first[h], last[h] = min(first[h], i), max(last[h], i)
    for h in first:
        n = sum(1 for l in lines[first[h]:last[h] + 1] if l.strip() and l.strip()[0] not in ';.' and not l.strip().endswith(':'))
        if n > 1500:
            print(h, first[h] + 1, last[h] + 1, n, "instructions")


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "spills":
        spills(sys.argv[2])
    elif cmd == "func":
        func(sys.argv[2], sys.argv[3])
    elif cmd == "hist":
        hist(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    elif cmd == "loops":
        loops(sys.argv[2])
    else:
        print(__doc__)
