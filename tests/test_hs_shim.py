"""The Haskell binding (haskell/Sonic/HIP.hs) against the header it binds (include/sonic_hip.h), without a Haskell toolchain.

SURVEY.md 8(f4): the image has no ghc / cabal / stack, so the module cannot be compiled here.  What CAN be checked mechanically is
the part a compiler would not check either -- `foreign import ccall` trusts the programmer: that every imported symbol exists in
the header, takes as many arguments as the Haskell type says, and that every argument and the result have the same machine kind
(pointer / 64-bit integer / C int / size_t / void).  The reference surface the module re-creates: /root/reference/sonic.cabal:31-37,
src/Sonic/Protocol.hs:28-52, src/Sonic/Signature.hs:22-37, src/Sonic/CommitmentScheme.hs:20-57, src/Sonic/SRS.hs:11-28.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "haskell", "Sonic", "HIP.hs")
HDR = os.path.join(ROOT, "include", "sonic_hip.h")


def c_kind(decl: str) -> str:
    d = decl.strip()
    if "*" in d or "[" in d:
        return "ptr"
    toks = [t for t in re.split(r"\s+", d) if t not in ("const", "unsigned")]
    base = toks[0]
    return {"int64_t": "i64", "int": "int", "size_t": "size", "void": "void", "uint32_t": "u32", "double": "f64"}[base]


def header_prototypes():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"\b(int|void|size_t|int64_t)\s+(sonic_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        kinds = [] if args in ("", "void") else [c_kind(a) for a in args.split(",")]
        protos[name] = (c_kind(ret), kinds)
    return protos


def split_arrows(t: str):
    parts, depth, cur = [], 0, ""
    i = 0
    while i < len(t):
        ch = t[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if depth == 0 and t.startswith("->", i):
            parts.append(cur.strip())
            cur = ""
            i += 2
            continue
        cur += ch
        i += 1
    parts.append(cur.strip())
    return parts


def hs_kind(t: str) -> str:
    t = t.strip()
    if t.startswith(("Ptr ", "FunPtr ", "CString")) or t == "CString":
        return "ptr"
    return {"Int64": "i64", "CInt": "int", "CSize": "size", "()": "void"}[t]


def hs_imports():
    src = open(HS).read()
    src = re.sub(r"--.*$", "", src, flags=re.M)
    out = []
    for m in re.finditer(r'foreign import ccall\s+(safe|unsafe)\s+"(&?)(\w+)"\s+(\w+)\s*::\s*(.*)', src):
        safety, addr, sym, hsname, ty = m.groups()
        out.append((safety, addr == "&", sym, hsname, ty.strip()))
    return out, src


def test_the_shim_exists_and_imports_something():
    imps, _ = hs_imports()
    assert len(imps) >= 30
    syms = {i[2] for i in imps}
    # the functions VERDICT r05 named, and the four modules' surface
    for need in ("sonic_srs_new", "sonic_srs_pairing", "sonic_commit_poly", "sonic_open_poly", "sonic_pc_v", "sonic_prove", "sonic_verify",
                 "sonic_hsc_prove_poly", "sonic_hsc_verify_poly", "sonic_prove_shared", "sonic_prove_batch", "sonic_srs_free", "sonic_prover_free"):
        assert need in syms, need


def test_every_foreign_import_matches_the_header():
    protos = header_prototypes()
    imps, _ = hs_imports()
    for safety, is_addr, sym, hsname, ty in imps:
        assert sym in protos, f"{hsname}: {sym} is not declared in include/sonic_hip.h"
        ret, args = protos[sym]
        if is_addr:
            # "&f" :: FunPtr (Ptr X -> IO ())  -- a finalizer: one pointer in, nothing out
            m = re.fullmatch(r"FunPtr \((.*)\)", ty)
            assert m, f"{hsname}: an address import must be a FunPtr"
            parts = split_arrows(m.group(1))
            assert [hs_kind(p) for p in parts[:-1]] == args == ["ptr"], (hsname, parts, args)
            assert parts[-1] == "IO ()" and ret == "void", (hsname, parts[-1], ret)
            continue
        parts = split_arrows(ty)
        res = parts[-1]
        assert res.startswith("IO "), f"{hsname}: result must be in IO"
        hs_args = [hs_kind(p) for p in parts[:-1]]
        assert len(hs_args) == len(args), f"{hsname} ({sym}): {len(hs_args)} arguments in Haskell, {len(args)} in C"
        assert hs_args == args, f"{hsname} ({sym}): argument kinds {hs_args} != header's {args}"
        assert hs_kind(res[3:]) == ret, f"{hsname} ({sym}): result {res} != header's {ret}"
        # a call that launches kernels or waits for the GPU must not be `unsafe` (it would block the RTS's capability)
        if safety == "unsafe":
            assert sym in ("sonic_abi_version", "sonic_last_error", "sonic_srs_d", "sonic_srs_device", "sonic_proof_size", "sonic_hsc_proof_size"), sym


def test_every_import_is_used_and_the_abi_version_is_the_headers():
    imps, src = hs_imports()
    for _, _, _, hsname, _ in imps:
        assert len(re.findall(r"\b%s\b" % re.escape(hsname), src)) >= 2, f"{hsname} is imported and never used"
    hdr = open(HDR).read()
    ver = int(re.search(r"#define SONIC_ABI_VERSION (\d+)", hdr).group(1))
    assert int(re.search(r"abiExpected = (\d+)", src).group(1)) == ver


def test_the_exports_are_the_reference_surface():
    src = open(HS).read()
    head = src[src.index("module Sonic.HIP"):src.index(") where")]
    for name in ("SRS", "new", "srsD", "gNegativeX", "gPositiveX", "hNegativeX", "hPositiveX", "gNegativeAlphaX", "gPositiveAlphaX",
                 "hNegativeAlphaX", "hPositiveAlphaX", "srsPairing", "commitPoly", "openPoly", "pcV", "prove", "verify", "hscProve", "hscVerify"):
        assert re.search(r"\b%s\b" % name, head), name
        assert re.search(r"^%s\b.*::" % name, src, flags=re.M) or name == "SRS" or re.search(r"^%s, .*::" % name, src, flags=re.M) \
            or re.search(r"^\w+(, \w+)*, %s\b.*::" % name, src, flags=re.M), f"{name} has no type signature"
    # no helper is left as "the obvious marshalling helper": each one VERDICT r05 listed is defined
    for helper in ("withFr", "withFrs", "withTerms", "frFromBytes", "g1FromBytes", "decodeProof", "fq12FromBytes", "lastError"):
        assert re.search(r"^%s\b.* =" % helper, src, flags=re.M) or re.search(r"^%s\b[^\n]*\n  " % helper, src, flags=re.M), helper
    frag = open(os.path.join(ROOT, "haskell", "package.fragment.yaml")).read()
    assert "extra-libraries" in frag and "sonic_hip" in frag and "Sonic.HIP" in frag
