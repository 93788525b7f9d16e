"""CPU-only checks of the host mirror and of the C-ABI library: the shared object loads and exports
every symbol include/libdeflate_hip.h declares (no compute call is made), and argument validation
reproduces the reference's usage errors (tests/Test.lua:2290-2503) before any device work."""
import os
import re

import pytest

from ldtest import ROOT


def test_library_exports_every_declared_symbol():
    from libdeflate_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "libdeflate_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ldh_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(N.lib, name), name
    assert sorted(N.EXPORTS) == declared
    assert N.lib.ldh_compress_bound(0) % 16 == 0 and N.lib.ldh_compress_bound(131072) >= 131072 + 20


def test_usage_errors_match_reference_strings():
    from libdeflate_amd.LibDeflate import LibDeflate as LD, UsageError
    cases = [
        (lambda: LD.CompressDeflate(None), "Usage: LibDeflate:CompressDeflate(str, configs): 'str' - string expected got 'nil'."),
        (lambda: LD.CompressDeflate(b"", 1), "Usage: LibDeflate:CompressDeflate(str, configs): 'configs' - nil or table expected got 'number'."),
        (lambda: LD.CompressDeflate(b"", {"level": 10}), "Usage: LibDeflate:CompressDeflate(str, configs): 'configs' - unsupported 'level': 10."),
        (lambda: LD.CompressZlib(b"", {"strategy": "x"}), "Usage: LibDeflate:CompressZlib(str, configs): 'configs' - unsupported 'strategy': 'x'."),
        (lambda: LD.CompressZlib(b"", {"foo": 1}), "Usage: LibDeflate:CompressZlib(str, configs): 'configs' - unsupported table key in the configs: 'foo'."),
        (lambda: LD.CompressDeflateWithDict(b"", None), "Usage: LibDeflate:CompressDeflateWithDict(str, dictionary, configs): 'dictionary' - table expected got 'nil'."),
        (lambda: LD.DecompressDeflate(1), "Usage: LibDeflate:DecompressDeflate(str): 'str' - string expected got 'number'."),
        (lambda: LD.DecompressZlibWithDict(b"", {}), "Usage: LibDeflate:DecompressZlibWithDict(str, dictionary): 'dictionary' - table expected got 'table'."),
        (lambda: LD.Adler32(None), "Usage: LibDeflate:Adler32(str): 'str' - string expected got 'nil'."),
        (lambda: LD.EncodeForPrint(None), "Usage: LibDeflate:EncodeForPrint(str): 'str' - string expected got 'nil'."),
        (lambda: LD.DecodeForPrint(3), "Usage: LibDeflate:DecodeForPrint(str): 'str' - string expected got 'number'."),
        (lambda: LD.CreateDictionary(b"", 0, 1), "Usage: LibDeflate:CreateDictionary(str, strlen, adler32): 'str' - Empty string is not allowed."),
        (lambda: LD.CreateDictionary(b"a", 2, 1), None),
        (lambda: LD.CreateDictionary(b"a" * 32769, 32769, 1), "Usage: LibDeflate:CreateDictionary(str, strlen, adler32): 'str' - string longer than 32768 bytes is not allowed. Got 32769 bytes."),
    ]
    for fn, msg in cases:
        with pytest.raises(UsageError) as e:
            fn()
        if msg is not None:
            assert str(e.value) == msg


def test_product_has_no_cpu_fallback():
    # the product package never references the oracle
    pkg = os.path.join(ROOT, "libdeflate_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "ld_oracle" not in txt and "libld_oracle" not in txt, f
