"""Static check of julia/ExaAdmmHip.jl against include/exaadmm_hip.h.

No Julia runtime exists in the build image, so the binding cannot be executed here.  What CAN be
checked without running it is everything a silent ABI slip would consist of: the field order and
types of the two structs passed by reference, the symbol / argument count / C types of every
`ccall`, and the hand-decoded byte layout of `exaadmm_info`.  The header is parsed with a small
C-declaration reader; the Julia file with regular expressions over its (regular) source."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "exaadmm_hip.h")
JL = os.path.join(ROOT, "julia", "ExaAdmmHip.jl")


# ----------------------------------------------------------------------------- header side
def _strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def _norm_ctype(t):
    t = re.sub(r"\bconst\b", " ", t)
    t = re.sub(r"\s+", " ", t).strip()
    t = t.replace(" *", "*").replace("* ", "*")
    return t


def parse_header():
    src = _strip_comments(open(HDR).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"((?:const\s+)?\w+)\s+(.*)$", decl, flags=re.S)
            base, rest = mm.group(1), mm.group(2)
            for item in rest.split(","):
                item = item.strip()
                stars = item.count("*")
                name = item.replace("*", "").strip()
                fields.append((name, _norm_ctype(base) + "*" * stars))
        structs[m.group(2)] = fields
    funcs = {}
    for m in re.finditer(r"(?:^|;|\})\s*((?:const\s+)?\w+\s*\*?)\s*(exaadmm_\w+)\s*\(([^;{]*?)\)\s*(?=;)", src, flags=re.S):
        ret, name, args = _norm_ctype(m.group(1)), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\[\d*\]", "*", a)            # double out[4] -> pointer
                stars = a.count("*")
                toks = a.replace("*", " ").split()
                base = " ".join(t for t in toks[:-1]) if len(toks) > 1 else toks[0]
                alist.append(_norm_ctype(base) + "*" * stars)
        funcs[name] = (ret, alist)
    return structs, funcs


C2CTYPES = {"int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double, "int": C.c_int}


def layout(fields):
    """(offsets, size) of a C struct with natural alignment, via ctypes"""
    fl = [(n, C.c_void_p if t.endswith("*") else C2CTYPES[t]) for n, t in fields]
    S = type("S", (C.Structure,), {"_fields_": fl})
    return {n: getattr(S, n).offset for n, _ in fl}, C.sizeof(S)


# ----------------------------------------------------------------------------- Julia side
JL2C_SCALAR = {"Int64": "int64_t", "Int32": "int32_t", "Float64": "double", "Cint": "int", "Cdouble": "double"}


def julia_struct(src, name):
    m = re.search(r"struct\s+%s\b[^\n]*\n(.*?)\nend" % name, src, flags=re.S)
    assert m, "struct %s not found in the Julia shim" % name
    body = re.sub(r"#[^\n]*", "", m.group(1))
    out = []
    for f in re.split(r"[;\n]", body):
        f = f.strip()
        if not f:
            continue
        fname, ftype = (x.strip() for x in f.split("::"))
        out.append((fname, ftype))
    return out


def jl_field_to_c(t):
    m = re.match(r"Ptr\{(\w+)\}$", t)
    if m:
        return JL2C_SCALAR[m.group(1)] + "*"
    return JL2C_SCALAR[t]


def split_top(s):
    """split on commas at bracket depth 0"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls(src):
    """[(symbol, ret, [argtypes], n_actual_args)] of every ccall into libexaadmm_hip"""
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libexaadmm_hip\)", src):
        i = m.start() + len("ccall")
        depth, j = 0, i
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        parts = split_top(src[i + 1:j])
        ret, types = parts[1], parts[2]
        assert types.startswith("(") and types.endswith(")")
        tl = [t for t in split_top(types[1:-1]) if t]
        calls.append((m.group(1), ret, tl, len(parts) - 3))
    return calls


def compatible(jl, c):
    """is Julia ccall argument type `jl` a correct way to pass C parameter type `c`?"""
    table = {
        "Ptr{Cvoid}": {"exaadmm_handle*", "void*"},
        "Cint": {"int"},
        "Cdouble": {"double"},
        "Int64": {"int64_t"},
        "Ptr{Float64}": {"double*"},
        "Ref{Float64}": {"double*"},
        "Ref{Int64}": {"int64_t*"},
        "Ref{Cint}": {"int*"},
        "Ptr{Int64}": {"int64_t*"},
        "Ref{ExaadmmGrid}": {"exaadmm_grid*"},
        "Ref{ExaadmmParams}": {"exaadmm_params*"},
        "Ref{Ptr{Cvoid}}": {"exaadmm_handle**"},
        "Ptr{UInt8}": {"exaadmm_info*"},
        "Cstring": {"char*"},
    }
    return c in table.get(jl, set())


# ----------------------------------------------------------------------------- tests
@pytest.fixture(scope="module")
def pkg():
    import exaadmm_loader
    return exaadmm_loader.load_package()


@pytest.fixture(scope="module")
def parsed():
    structs, funcs = parse_header()
    src = open(JL).read()
    return structs, funcs, src


def test_header_parser_sees_the_whole_abi(parsed):
    structs, funcs, _ = parsed
    assert set(structs) >= {"exaadmm_grid", "exaadmm_params", "exaadmm_info"}
    assert len(funcs) >= 50 and "exaadmm_create" in funcs and "exaadmm_mp_create" in funcs
    assert funcs["exaadmm_update_residual"] == ("int", ["exaadmm_handle*", "double*"])
    assert funcs["exaadmm_version"] == ("char*", [])


@pytest.mark.parametrize("jl_name,c_name", [("ExaadmmGrid", "exaadmm_grid"), ("ExaadmmParams", "exaadmm_params")])
def test_struct_fields_match_in_order_and_type(parsed, jl_name, c_name):
    structs, _, src = parsed
    jf = julia_struct(src, jl_name)
    cf = structs[c_name]
    assert [n for n, _ in jf] == [n for n, _ in cf], "field order differs"
    for (n, jt), (_, ct) in zip(jf, cf):
        assert jl_field_to_c(jt) == ct, "field %s: Julia %s vs C %s" % (n, jt, ct)
    # Julia lays immutable / mutable structs of isbits fields out like C (natural alignment): same size
    offs, size = layout(cf)
    assert size % 8 == 0 and offs[cf[0][0]] == 0


def test_struct_constructors_pass_one_value_per_field(parsed):
    structs, _, src = parsed
    n_grid = len(structs["exaadmm_grid"])
    ctor_calls = 0
    for m in re.finditer(r"\bExaadmmGrid\(", src):
        if src[max(0, m.start() - 7):m.start()].endswith("struct "):
            continue
        i = m.end() - 1
        depth, j = 0, i
        while True:
            depth += src[j] == "("
            depth -= src[j] == ")"
            if depth == 0:
                break
            j += 1
        args = split_top(src[i + 1:j])
        assert len(args) == n_grid, "ExaadmmGrid(...) built with %d values for %d fields" % (len(args), n_grid)
        # positional order: every `pointer(d.X)` argument must sit at field X's position
        names = [n for n, _ in structs["exaadmm_grid"]]
        for pos, a in enumerate(args):
            mm = re.match(r"pointer\(d\.(\w+)\)$", a)
            if mm:
                assert names[pos] == mm.group(1), "argument %d is %s but the field there is %s" % (pos, a, names[pos])
        ctor_calls += 1
    assert ctor_calls >= 2
    m = re.search(r"to_params\(par::Parameters\)\s*=\s*ExaadmmParams\((.*?)\)\n", src, flags=re.S)
    args = split_top(m.group(1))
    pnames = [n for n, _ in structs["exaadmm_params"]]
    assert len(args) == len(pnames)
    for pos, a in enumerate(args):
        mm = re.match(r"par\.(\w+)$", a)
        if mm:
            assert pnames[pos] == mm.group(1)


def test_every_ccall_matches_a_prototype(parsed):
    _, funcs, src = parsed
    calls = julia_ccalls(src)
    assert len(calls) >= 20
    for sym, ret, types, nargs in calls:
        assert sym in funcs, "ccall of %s: no such entry point in include/exaadmm_hip.h" % sym
        cret, cargs = funcs[sym]
        assert compatible(ret, cret) or (ret, cret) == ("Cint", "int"), "%s: return %s vs %s" % (sym, ret, cret)
        assert len(types) == len(cargs), "%s: %d argument types for %d parameters" % (sym, len(types), len(cargs))
        assert nargs == len(types), "%s: %d values for %d argument types" % (sym, nargs, len(types))
        for k, (jt, ct) in enumerate(zip(types, cargs)):
            assert compatible(jt, ct), "%s argument %d: Julia %s cannot pass C %s" % (sym, k, jt, ct)


def test_info_byte_decode_matches_the_struct(parsed):
    structs, _, src = parsed
    fields = structs["exaadmm_info"]
    offs, size = layout(fields)
    m = re.search(r"info = zeros\(UInt8, (\d+)\)", src)
    assert int(m.group(1)) >= size, "info buffer smaller than sizeof(exaadmm_info) = %d" % size
    # ints = reinterpret(Int32, info[1:16]); dbl = reinterpret(Float64, info[17:end])
    assert re.search(r"ints = reinterpret\(Int32, info\[1:16\]\)", src) and re.search(r"dbl = reinterpret\(Float64, info\[17:end\]\)", src)
    int_names = [n for n, t in fields if t == "int32_t"]
    dbl_names = [n for n, t in fields if t == "double"]
    assert [offs[n] for n in int_names] == [0, 4, 8, 12] and int_names == ["status", "inner", "outer", "cumul"]
    assert [offs[n] for n in dbl_names] == [16 + 8 * k for k in range(len(dbl_names))]
    # the decode reads ints[1..4] as status, inner, outer, cumul and dbl[1..8] as the first eight doubles
    assert re.search(r"ints\[1\] == 1 \? :Solved", src)
    assert re.search(r"mod\.info\.inner, mod\.info\.outer, mod\.info\.cumul = ints\[2\], ints\[3\], ints\[4\]", src)
    assert re.search(r"mod\.info\.objval, mod\.info\.primres, mod\.info\.dualres, mod\.info\.mismatch = dbl\[1:4\]", src)
    assert re.search(r"mod\.info\.norm_z_curr, mod\.info\.norm_z_prev, mod\.info\.eps_pri, mod\.info\.time_overall = dbl\[5:8\]", src)
    assert dbl_names[:8] == ["objval", "primres", "dualres", "mismatch", "norm_z_curr", "norm_z_prev", "eps_pri", "time_overall"]


def test_python_mirror_structs_agree_with_the_header(parsed, pkg):
    """the ctypes structs the GPU tests go through are checked the same way"""
    structs, _, _ = parsed
    for cname, cls in (("exaadmm_grid", pkg.Grid), ("exaadmm_params", pkg.Parameters), ("exaadmm_info", pkg.Info)):
        offs, size = layout(structs[cname])
        assert C.sizeof(cls) == size, cname
        for (hn, _), (pn, _) in zip(structs[cname], cls._fields_):
            py_off = getattr(cls, pn).offset
            assert py_off == offs[hn], "%s.%s at %d, header says %d" % (cname, pn, py_off, offs[hn])


def test_vector_ids_match(parsed):
    _, _, src = parsed
    hdr = _strip_comments(open(HDR).read())
    ids = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"EXAADMM_(\w+)\s*=\s*(\d+)", hdr))
    m = re.search(r"const (U_CURR.*?) = 0:9", src)
    names = [x.strip() for x in m.group(1).split(",")]
    assert [ids[n] for n in names] == list(range(10))
