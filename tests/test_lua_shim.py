"""The LuaJIT shim cannot run in the build image (no Lua VM), so what can be checked statically is:
every prototype in its ffi.cdef blocks equals the declaration in include/libdeflate_hip.h token for token, the method
table has exactly the reference's 22 keys (tests/Test.lua:2712-2735), `internals` its six (LibDeflate.lua:3426-3433),
and the batched entry point is a module of its own that `require` can reach."""
import os
import re

from ldtest import ROOT

LUA = os.path.join(ROOT, "lua")


def _tokens(text):
    return re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*|\(|\)|,|\[|\]|;", text)


def _prototypes(text):
    """{function name: token list} of every `... ldh_xxx(...);` in a piece of C."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*\n]*?\b(ldh_[a-z0-9_]+)\s*\([^;{]*?\))\s*;", text):
        out[m.group(2)] = _tokens(m.group(1))
    return out


def _cdefs(path):
    src = open(path).read()
    blocks = re.findall(r"ffi\.cdef\s*\[\[(.*?)\]\]", src, flags=re.S)
    assert blocks, path
    return "\n".join(blocks)


def test_cdef_prototypes_equal_the_header():
    header = _prototypes(open(os.path.join(ROOT, "include", "libdeflate_hip.h")).read())
    assert len(header) > 30
    seen = set()
    for f in ("LibDeflate.lua", "LibDeflateBatch.lua"):
        protos = _prototypes(_cdefs(os.path.join(LUA, f)))
        assert protos, f
        for name, toks in protos.items():
            assert name in header, "%s: %s is not declared in the header" % (f, name)
            assert toks == header[name], "%s: prototype of %s differs from the header" % (f, name)
            seen.add(name)
    # every entry point the shims call is declared in one of their cdefs
    called = set()
    for f in ("LibDeflate.lua", "LibDeflateBatch.lua"):
        called |= set(re.findall(r"\bC\.(ldh_[a-z0-9_]+)", open(os.path.join(LUA, f)).read()))
    assert called <= seen, called - seen
    for must in ("ldh_compress", "ldh_decompress", "ldh_adler32", "ldh_print6_encode", "ldh_print6_decode",
                 "ldh_dict_create", "ldh_compress_batch", "ldh_decompress_batch", "ldh_adler32_batch",
                 "ldh_print6_encode_batch", "ldh_print6_decode_batch"):
        assert must in called, must


def test_method_table_has_exactly_the_22_reference_keys():
    src = open(os.path.join(LUA, "LibDeflate.lua")).read()
    want = {"EncodeForWoWChatChannel", "_COPYRIGHT", "DecodeForWoWAddonChannel", "CompressDeflate", "DecompressDeflate",
            "CompressDeflateWithDict", "DecompressZlibWithDict", "CreateCodec", "DecodeForWoWChatChannel", "internals",
            "_VERSION", "_MAJOR", "_MINOR", "Adler32", "CreateDictionary", "CompressZlibWithDict", "EncodeForPrint",
            "CompressZlib", "DecodeForPrint", "DecompressDeflateWithDict", "EncodeForWoWAddonChannel", "DecompressZlib"}
    keys = set(re.findall(r"function LibDeflate:([A-Za-z0-9_]+)", src))
    keys |= set(re.findall(r"\bLibDeflate\.([A-Za-z_][A-Za-z0-9_]*)\s*=", src))
    keys |= set(re.findall(r'^wrap\("([A-Za-z]+)"', src, flags=re.M))
    keys |= set(re.findall(r"\b(_[A-Z]+)\s*=", src[src.index("local LibDeflate = {"): src.index("local STRATEGY")]))
    keys |= set(re.findall(r"\b((?:En|De)codeForWoW[A-Za-z]+)\s*=\s*\{", src))
    assert keys == want, (keys - want, want - keys)
    internals = src[src.index("LibDeflate.internals = {"):]
    internals = internals[: internals.index("}")]
    assert set(re.findall(r"^\s*([A-Za-z0-9_]+)\s*=", internals, flags=re.M)) == {
        "LoadStringToTable", "IsValidDictionary", "IsEqualAdler32", "_byte_to_6bit_char", "_6bit_to_byte", "InternalClearCache"}
    # one return value (require drops a second one), no globals assigned at top level
    assert re.search(r"^return LibDeflate\s*$", src, flags=re.M)
    assert not re.search(r"^return LibDeflate\s*,", src, flags=re.M)


def test_batch_module_is_reachable_and_complete():
    src = open(os.path.join(LUA, "LibDeflateBatch.lua")).read()
    assert set(re.findall(r"^function Batch\.([A-Za-z0-9]+)", src, flags=re.M)) >= {
        "Compress", "Decompress", "Adler32", "EncodeForPrint", "DecodeForPrint", "CompressBuffers", "DecompressBuffers",
        "Alloc", "AllocPinned", "CompressBound"}
    assert re.search(r"^return Batch\s*$", src, flags=re.M)
    assert 'require("LibDeflate")' in src and 'package.loaded["LibDeflate.__hip"]' in src
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert 'require("LibDeflateBatch")' in integ
    assert "local LibDeflate, Batch = require" not in integ


def _call_args(src, fn, callee):
    """argument expressions of the `callee(...)` call inside `function Batch.<fn>`"""
    body = src[src.index("function Batch.%s(" % fn):]
    body = body[: body.index("\nend\n")]
    i = body.index(callee + "(") + len(callee) + 1
    depth, args, cur = 1, [], ""
    while depth:
        ch = body[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if not depth:
                break
        if ch == "," and depth == 1:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    args.append(cur.strip())
    return args


def test_buffer_forms_hand_the_callers_pointers_to_the_c_abi():
    # a batched Lua entry point that can carry a real batch -- caller-owned cdata, no Lua string, no
    # ffi.new of the batch's size.  Statically: the argument lists line up with the header's prototypes, each pointer
    # is cast to the parameter's own type, and no allocation of the batch happens inside
    src = open(os.path.join(LUA, "LibDeflateBatch.lua")).read()
    header = open(os.path.join(ROOT, "include", "libdeflate_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    for fn, callee in (("CompressBuffers", "ldh_compress_batch"), ("DecompressBuffers", "ldh_decompress_batch")):
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % callee, header)
        params = [p.strip() for p in m.group(1).split(",")]
        args = _call_args(src, fn, "C." + callee)
        assert len(args) == len(params), (fn, args, params)
        for a, p in zip(args, params):
            if "*" in p and "ldh_dict" not in p:
                ctype = p[: p.rindex("*") + 1].strip()
                assert a.startswith('ffi.cast("%s"' % ctype), (fn, a, p)
        body = src[src.index("function Batch.%s(" % fn):]
        body = body[: body.index("\nend\n")]
        assert "ffi.new" not in body and "table.concat" not in body and "ffi.string" not in body, fn
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "CompressBuffers" in integ and "DecompressBuffers" in integ
