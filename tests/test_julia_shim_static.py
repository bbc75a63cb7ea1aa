"""CPU: the Julia shim (julia/MCMCDiagnosticToolsROC.jl) cannot be executed in this image -- there is no Julia -- so what can be
held to the C-ABI without running it is held here, by reading its text: every `ccall` names a symbol include/mdt_hip.h declares,
passes as many arguments as the prototype takes (and as many values as it lists types), and `struct MdtArgs` has the fields of
`mdt_args` in the header's order with types of the same size.  A static check, not a run: runtests.jl stays the real test."""
import os
import re

import __graft_entry__ as g

ROOT = g.ROOT
JL = os.path.join(ROOT, "julia", "MCMCDiagnosticToolsROC.jl")
HDR = os.path.join(ROOT, "include", "mdt_hip.h")


def _strip_c_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _split_top_level(s):
    """split at commas outside brackets"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(text, start):
    """text[start] == '(' -> index behind its closing bracket"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced brackets")


def header_prototypes():
    src = _strip_c_comments(open(HDR).read())
    protos = {}
    for m in re.finditer(r"\b(mdt_[a-z_]+)\s*\(", src):
        end = _balanced(src, m.end() - 1)
        params = src[m.end():end - 1].strip()
        protos[m.group(1)] = 0 if params in ("", "void") else len(_split_top_level(params))
    return protos


def julia_ccalls():
    src = re.sub(r"#.*", "", open(JL).read())  # (no '#' inside the strings of the shim's ccall sites)
    calls = []
    for m in re.finditer(r"ccall\(\(:(mdt_[a-z_]+),\s*libmdt\)", src):
        start = src.rfind("ccall(", 0, m.end())
        end = _balanced(src, start + len("ccall"))
        args = _split_top_level(src[start + len("ccall("):end - 1])
        # args: (sym, lib), rettype, (argtypes...), values...
        types = args[2].strip()
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        inner = types[1:-1].strip()
        tl = [t for t in _split_top_level(inner) if t]
        calls.append((m.group(1), args[1], tl, args[3:]))
    return calls


def test_every_ccall_names_a_declared_symbol_with_the_prototypes_arity():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 12
    for name, ret, types, values in calls:
        assert name in protos, f"{name}: not declared in include/mdt_hip.h"
        assert len(types) == protos[name], f"{name}: {len(types)} argument types in the shim, {protos[name]} parameters in the header"
        assert len(values) == len(types), f"{name}: {len(values)} values for {len(types)} types"
        assert ret in ("Cint", "Cstring"), (name, ret)


def header_param_types():
    src = _strip_c_comments(open(HDR).read())
    out = {}
    for m in re.finditer(r"\b(mdt_[a-z_]+)\s*\(", src):
        end = _balanced(src, m.end() - 1)
        params = re.sub(r"\s+", " ", src[m.end():end - 1]).strip()
        out[m.group(1)] = [] if params in ("", "void") else [re.sub(r"\s*[A-Za-z_0-9]+$", "", p_).strip() for p_ in _split_top_level(params)]
    return out


def _c_class(t):
    if "*" in t:
        return "ptr"
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64"}[t.replace("const ", "")]


def _jl_class(t):
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cdouble": "f64", "Float64": "f64"}[t]


def test_every_ccall_passes_the_prototypes_argument_classes():
    """pointer / 32-bit integer / 64-bit integer / double, position by position: what the x86-64 calling convention cares about"""
    ctypes_ = header_param_types()
    for name, _, types, _ in julia_ccalls():
        want = [_c_class(t) for t in ctypes_[name]]
        got = [_jl_class(t) for t in types]
        assert got == want, (name, types, ctypes_[name])
        # a struct goes by reference as Ref{MdtArgs}, never by value
        for jt, ct in zip(types, ctypes_[name]):
            if "mdt_args" in ct:
                assert jt == "Ref{MdtArgs}", (name, jt)


def test_the_hot_path_entry_points_are_bound():
    bound = {c[0] for c in julia_ccalls()}
    for name in ("mdt_ctx_create", "mdt_ctx_create_multi", "mdt_ctx_status", "mdt_ctx_reserve", "mdt_last_error", "mdt_shard_bounds",
                 "mdt_ess_rhat", "mdt_bulk_tail", "mdt_rhat_nested", "mdt_mcse_nested", "mdt_mcse_mean", "mdt_mcse_std", "mdt_mcse_quantile",
                 "mdt_bfmi", "mdt_gelmandiag"):
        assert name in bound, name


C_SIZES = {"int64_t": 8, "int32_t": 4, "double": 8, "void*": 8, "const void* const*": 8}
JL_SIZES = {"Int64": 8, "Int32": 4, "Float64": 8}


def test_mdtargs_mirrors_the_header_struct_field_by_field():
    hdr = _strip_c_comments(open(HDR).read())
    body = re.search(r"typedef struct mdt_args \{(.*?)\} mdt_args;", hdr, flags=re.S).group(1)
    cfields = []
    for decl in [d.strip() for d in body.split(";") if d.strip()]:
        m = re.match(r"(.*?[\s\*])([a-z_0-9]+(?:\s*,\s*[a-z_0-9]+)*)$", decl, flags=re.S)
        ctype = re.sub(r"\s+", " ", m.group(1)).strip().replace(" *", "*")
        for name in [n.strip() for n in m.group(2).split(",")]:
            cfields.append((name, C_SIZES[ctype]))
    jl = re.sub(r"#.*", "", open(JL).read())
    jbody = re.search(r"struct MdtArgs\n(.*?)\nend", jl, flags=re.S).group(1)
    jfields = []
    for line in [l.strip() for l in jbody.splitlines() if l.strip()]:
        name, jtype = [t.strip() for t in line.split("::")]
        jfields.append((name, 8 if jtype.startswith("Ptr") else JL_SIZES[jtype]))
    assert [n for n, _ in jfields] == [n for n, _ in cfields]
    assert [s for _, s in jfields] == [s for _, s in cfields]
    # the same 120 bytes tests/test_capi_symbols.py holds the ctypes mirror to (both layouts have no padding: 8-byte fields on 8-byte offsets)
    off = 0
    for _, size in cfields:
        assert off % size == 0
        off += size
    assert off == 120


def _julia_code_only(src):
    """comments, strings and character literals blanked out"""
    out, i, n = [], 0, len(src)
    while i < n:
        if src.startswith('"""', i):
            j = src.index('"""', i + 3)
            i = j + 3
        elif src[i] == '"':
            j = i + 1
            while src[j] != '"':
                if src[j] == "\\":
                    j += 1
                j += 1
            out.append('""')
            i = j + 1
        elif src[i] == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src[i] == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src[i + 3] == "'")):
            i = src.index("'", i + 2 if src[i + 1] != "\\" else i + 3) + 1
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


def test_blocks_and_brackets_balance_in_the_julia_sources():
    """a crude stand-in for the parser this image lacks: outside brackets (generators, comprehensions and a[end] live inside them)
    every block opener has its `end`, and every bracket closes -- an edit that drops an `end` fails here, not on a user's machine"""
    for rel in ("MCMCDiagnosticToolsROC.jl", os.path.join("test", "runtests.jl"), "gen_reference_golden.jl"):
        code = _julia_code_only(open(os.path.join(ROOT, "julia", rel)).read())
        depth, opens, ends, curly = 0, 0, 0, 0
        for m in re.finditer(r"[\[\]\(\)\{\}]|\b(function|if|for|while|let|try|struct|type|module|begin|do|quote|macro|end)\b", code):
            t = m.group(0)
            if t in "[(":
                depth += 1
            elif t in "])":
                depth -= 1
                assert depth >= 0, rel
            elif t == "{":
                curly += 1
            elif t == "}":
                curly -= 1
                assert curly >= 0, rel
            elif depth == 0:
                if t == "end":
                    ends += 1
                else:
                    opens += 1
        assert depth == 0 and curly == 0, rel
        assert opens == ends, (rel, opens, ends)


def test_the_shim_exports_what_the_python_mirror_offers():
    """the two host mirrors offer the same requests (the Python one is the tested one): every request function of the package is
    defined and exported by the shim (`rank_normalize` is the Python side's stage output for parity checks, `from_mcmcchains_layout`
    is `samples_from_chains` there: a view, not a copy)"""
    jl = _julia_code_only(open(JL).read())
    exported = set()
    for m in re.finditer(r"^export (.*)$", jl, flags=re.M):
        exported |= {t.strip() for t in m.group(1).split(",")}
    for name in ("ess", "rhat", "ess_rhat", "bulk_tail", "rhat_nested", "mcse", "mcse_rhat_nested", "bfmi", "gelmandiag", "status",
                 "reserve", "shard_bounds", "AutocovMethod", "FFTAutocovMethod", "BDAAutocovMethod"):
        assert name in exported, name
        assert re.search(r"^(function |struct |abstract type )?" + name + r"\b", jl, flags=re.M), name
    pkg = g.load_package()
    for name in ("ess", "rhat", "ess_rhat", "bulk_tail", "rhat_nested", "mcse", "mcse_rhat_nested", "bfmi", "gelmandiag", "status", "reserve", "shard_bounds"):
        assert callable(getattr(pkg, name)), name
