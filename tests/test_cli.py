"""Command line mirror: messages and exit codes of tests/Test.lua:2537-2690."""
import io
import os

import pytest

from libdeflate_amd import cli

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run(args):
    out, err = io.StringIO(), io.StringIO()
    rc = cli.main(args.split() if isinstance(args, str) else args, out, err)
    return rc, out.getvalue(), err.getvalue()


def test_help_and_version():
    rc, out, err = run("-h")
    assert rc == 0 and err == ""
    for line in ("[OPTION] [INPUT] [OUTPUT]\n", "  -0    store only. no compression.\n", "  -1    fastest compression.\n",
                 "  -9    slowest and best compression.\n", "  -d    do decompression instead of compression.\n",
                 "  --dict <filename> specify the file that contains the entire preset dictionary.\n",
                 "  -h    give this help.\n",
                 "  --strategy <fixed/huffman_only/dynamic> specify a special compression strategy.\n",
                 "  -v    print the version and copyright info.\n",
                 "  --zlib  use zlib format instead of raw deflate.\n"):
        assert line in out
    rc, out, err = run("-v")
    assert rc == 0 and err == "" and "LibDeflate 1.0.2-release" in out


def test_errors(tmp_path):  # T:2580-2625 (everything here fails before any compute: no GPU needed)
    item = os.path.join(GOLD, "item_strings.txt")
    rc, out, err = run("-invalid")
    assert rc != 0 and out == "" and "LibDeflate: Invalid argument: -invalid" in err
    rc, out, err = run([item, "--dict"])
    assert rc != 0 and out == "" and "You must speicify the dict filename" in err
    rc, out, err = run([item, "--dict", "DNE"])
    assert rc != 0 and out == "" and "LibDeflate: Cannot read the dictionary file 'DNE':" in err
    rc, out, err = run("DNE DNE")
    assert rc != 0 and out == "" and "LibDeflate: Cannot read the file 'DNE':" in err
    rc, out, err = run([item, ".."])
    assert rc != 0 and out == "" and "LibDeflate: Cannot write the file '..':" in err
    rc, out, err = run([item])
    assert rc != 0 and out == "" and "LibDeflate: You must specify both input and output files." in err


@pytest.mark.gpu
def test_compress_and_decompress(tmp_path):  # T:2627-2690
    from libdeflate_amd import LibDeflate as LD
    g = lambda n: os.path.join(GOLD, n)  # noqa: E731
    dct = open(g("dictionary32768.txt"), "rb").read()
    d = LD.CreateDictionary(dct, len(dct), LD.Adler32(dct))
    funcs = ["CompressDeflate", "CompressDeflateWithDict", "CompressZlib", "CompressZlibWithDict", "DecompressDeflate",
             "DecompressDeflateWithDict", "DecompressZlib", "DecompressZlibWithDict"]
    args = ["", "--dict " + g("dictionary32768.txt"), "--zlib", "--zlib --dict " + g("dictionary32768.txt"), "-d",
            "-d --dict " + g("dictionary32768.txt"), "-d --zlib", "-d --zlib --dict " + g("dictionary32768.txt")]
    inputs = [g("item_strings.txt")] * 4 + [g("item_strings_deflate.bin"), g("item_strings_deflate_with_dict.bin"),
                                           g("item_strings_zlib.bin"), g("item_strings_zlib_with_dict.bin")]
    extra = [("-0", {"level": 0}), ("-1 --strategy huffman_only", {"level": 1, "strategy": "huffman_only"}),
             ("-5 --strategy dynamic", {"level": 5, "strategy": "dynamic"}),
             ("-9 --strategy fixed", {"level": 9, "strategy": "fixed"}), ("", None)]
    tmp = str(tmp_path / "cli.tmp")
    for k, fn in enumerate(funcs):
        for a, cfg in extra:
            rc, out, err = run((args[k] + " " + a).split() + [inputs[k], tmp])
            got = open(tmp, "rb").read()
            assert rc == 0 and out == "" and ("Successfully writes %d bytes" % len(got)) in err
            data = open(inputs[k], "rb").read()
            call = [data] + ([d] if "Dict" in fn else []) + ([cfg] if fn.startswith("Compress") else [])
            assert got == getattr(LD, fn)(*call)[0], (fn, a)
    # a text file is no deflate stream
    rc, out, err = run(["-d", g("item_strings.txt"), tmp])
    assert rc != 0 and out == "" and "LibDeflate: Decompress fails." in err
