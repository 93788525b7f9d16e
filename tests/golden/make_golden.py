#!/usr/bin/env python3
"""Regenerates tests/golden/ from the upstream tree (run once, in the build container).

Fixtures are DATA only: files that the reference's own tests hold
(/root/reference/tests/data/..., tests/dictionary32768.txt) are copied verbatim, and
the known-answer vectors of tests/Test.lua (inflate vectors :1311-1634, Adler-32 KATs
:1707-1739, pinned sizes :2210-2245 / :2694-2707, benchmark ratios docs/benchmark.md)
are transcribed into kat.json as hex/expected pairs.  No reference source text is kept.
"""
import json, os, shutil, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

COPY = [
    ("tests/data/reference/item_strings.txt", "item_strings.txt"),
    ("tests/data/reference/item_strings_deflate.txt", "item_strings_deflate.bin"),
    ("tests/data/reference/item_strings_deflate_with_dict.txt", "item_strings_deflate_with_dict.bin"),
    ("tests/data/reference/item_strings_zlib.txt", "item_strings_zlib.bin"),
    ("tests/data/reference/item_strings_zlib_with_dict.txt", "item_strings_zlib_with_dict.bin"),
    ("tests/data/reference/decode_6bit_weakaura.txt", "decode_6bit_weakaura.bin"),
    ("tests/data/reference/encode_6bit_weakaura.txt", "encode_6bit_weakaura.txt"),  # == data/warlockWeakAuras.txt
    ("tests/dictionary32768.txt", "dictionary32768.txt"),
    ("tests/dict_for_wow.txt", "dict_for_wow.txt"),
    ("tests/data/smalltest_no_newline.txt", "smalltest_no_newline.txt"),
    ("tests/data/adler32Test.txt", "adler32Test.txt"),
    ("tests/data/adler32Test2.txt", "adler32Test2.txt"),
    ("tests/data/totalrp3.txt", "totalrp3.txt"),
    ("tests/data/64kminus1.txt", "64kminus1.txt"),
    ("tests/data/64k.txt", "64k.txt"),
    ("tests/data/64kplus1.txt", "64kplus1.txt"),
]

def h(s):  # "4 0 fe ff" -> canonical hex
    return "".join("%02x" % int(t, 16) for t in s.split())

A65535 = "<61*65535>"  # run-length notation: <hex*count>, expanded by tests/ldtest.py:kat_hex

# (compressed hex, expected-output hex or None for nil, is_zlib)  -- Test.lua:1311-1550
INFLATE = [
    ("0100 00ff ff".replace(" ", ""), "", 0),
    (h("01 01 00 fe ff 0a"), "0a", 0),
    (h("01 ff ff 00 00") + A65535, A65535, 0),
    (h("00 ff ff 00 00") + A65535 + h("01 ff ff 00 00") + A65535, A65535 * 2, 0),
    (h("01 01 00 fe ff 31"), "31", 0),
    (h("03 00"), "", 0),
    (h("33 04 00"), "31", 0),
    (h("32 04 00 ff ff 00 00") + A65535 + h("01 ff ff 00 00") + A65535, "31" + A65535 * 2, 0),
    (h("01 01 00 fe ff 0a 00"), "0a", 0),
    (h("01 01 00 fe ff"), None, 0),
    (h("01 02 00 fd ff 01"), None, 0),
    ("", None, 0),
    (h("00"), None, 0), (h("02"), None, 0), (h("04"), None, 0),
    (h("00 01 00 fe ff"), None, 0),
    (h("04 80 49 92 24 49 92 24 0f b4 ff ff c3 04"), None, 0),
    (h("0c c0 81 00 00 00 00 00 90 ff 6b 04"), None, 0),
    (h("1a 07"), None, 0),
    (h("02 7e ff ff"), None, 0),
    (h("04 c0 81 08 00 00 00 00 20 7f eb 0b 00 00"), None, 0),
    (h("04 00 24 e9 ff 6d"), None, 0),
    (h("04 80 49 92 24 49 92 24 0f b4 ff ff c3 84"), None, 0),
    (h("04 80 49 92 24 49 92 24 71 ff ff 93 11 00"), None, 0),
    (h("04 00 24 e9 ff ff"), None, 0),
    (h("04 00 24 49"), None, 0),
    (h("04 00 fe ff"), None, 0),
    (h("fc 00 00"), None, 0),
    (h("00 00 00 00 00"), None, 0),
    (h("63 00"), None, 0),
    (h("63 18 05"), None, 0),
    (h("63 18 68 30 d0 0 0"), "<00*257>", 0),
    (h("3 00"), "", 0),
    ("", None, 1),
    (h("77 85"), None, 1), (h("70 85"), None, 1), (h("88 9c"), None, 1), (h("f8 9c"), None, 1),
    (h("78 90"), None, 1),
    (h("78 9c 63 00 00 00 01 00 01"), "00", 1),
    (h("78 9c 63 00 00 00 01 00"), None, 1),
    (h("78 9c 63 00 00 00 01 00 02"), None, 1),
    (h("78 9c 63 0"), None, 1),
    (h("6"), None, 0),
    (h("1 1 0 fe ff 0"), "00", 0),
    (h("4 0 24 49 0"), None, 0),
    (h("4 80 49 92 24 49 92 24 f b4 ff ff c3 84"), None, 0),
    (h("c c0 81 0 0 0 0 0 90 ff 6b 4 0"), None, 0),
    (h("1f 8b 8 0 0 0 0 0 0 0 3 0 0 0 0 1"), None, 0),
    (h("1f 8b 8 0 0 0 0 0 0 0 3 0 0 0 0 0 0 0 0 1"), None, 0),
    (h("5 c0 21 d 0 0 0 80 b0 fe 6d 2f 91 6c"), "", 0),
    (h("05 e0 81 91 24 cb b2 2c 49 e2 0f 2e 8b 9a 47 56 9f fb fe ec d2 ff 1f"), "", 0),
    (h("ed c0 1 1 0 0 0 40 20 ff 57 1b 42 2c 4f"), "<00*516>", 0),
    (h("ed cf c1 b1 2c 47 10 c4 30 fa 6f 35 1d 1 82 59 3d fb be 2e 2a fc f c"), "<00*518>", 0),
    (h("ed c0 81 0 0 0 0 80 a0 fd a9 17 a9 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0"), None, 0),
    (h("2 8 20 80 0 3 0"), "", 0),
    (h("63 18 5 40 c 0"), "<00*262>", 0),
    (h("e5 e0 81 ad 6d cb b2 2c c9 01 1e 59 63 ae 7d ee fb 4d fd b5 35 41 68"), None, 0),
    (h("25 fd 81 b5 6d 59 b6 6a 49 ea af 35 6 34 eb 8c b9 f6 b9 1e ef 67 49"), None, 0),
    (h("3 7e 0 0 0 0 0"), None, 0),
    (h("1b 7 0 0 0 0 0"), None, 0),
    (h("d c7 1 ae eb 38 c 4 41 a0 87 72 de df fb 1f b8 36 b1 38 5d ff ff 0"), None, 0),
    (h("63 18 5 8c 10 8 0 0 0 0"), "<00*258>" + "0001" * 4, 0),
    (h("63 60 60 18 c9 0 8 18 18 18 26 c0 28 0 29 0 0 0"), "<00*261>" + "90" + "00" * 6 + "9000", 0),
    (h("63 0 3 0 0 0 0 0"), "00" * 6, 0),
    (h("78"), None, 1),
    (h("1"), None, 0),
    (h("1 1 0"), None, 0),
    (h("1 1 0 ff ff 0"), None, 0),
    (h("1 1 0 fe fe 0"), None, 0),
    (h("1 34 43 cb bc") + "<00*17204>", "<00*17204>", 0),
    (h("1 34 43 cb bc") + "<00*17203>", None, 0),
    (h("1 34 43 cb bc") + "<00*17202>", None, 0),
] + [("%02x" % i, None, 0) for i in range(256)]

# Test.lua:1602-1634 (dictionary "abcdefgh"): (hex, expected hex or None, use_dict)
INFLATE_DICT = [
    (h("78 9c 63 00 00 00 01 00 01"), "00", 0),
    (h("78 bb 63 00 00 00 01 00 01"), None, 0),
    (h("78 bb 25 03 00 0e 63 00 00 00 01 00 01"), None, 0),
    (h("78 bb 0e 00 03 25 63 00 00 00 01 00 01"), "00", 1),
    (h("78 bb 0e 00 03"), None, 1), (h("78 bb 0e 00"), None, 1), (h("78 bb 0e"), None, 1),
    (h("78 bb"), None, 1),
    (h("78 bb 25 03 00 0e 63 00 00 00 01 00 01"), None, 1),
    (h("78 bb 0e 00 03 26 63 00 00 00 01 00 01"), None, 1),
]

ADLER = [  # Test.lua:1707-1735, LibDeflate.lua:473, examples/example.lua:100
    ("", 1), ("1", 0x00320032), ("12", 0x00960064), ("123", 0x012D0097), ("1234", 0x01F800CB),
    ("12345", 0x02F80100), ("123456", 0x042E0136), ("1234567", 0x059B016D), ("12345678", 0x074001A5),
    ("123456789", 0x091E01DE), ("1234567890", 0x0B2C020E), ("1234567890a", 0x0D9B026F),
    ("1234567890ab", 0x106C02D1), ("1234567890abc", 0x13A00334), ("1234567890abcd", 0x17380398),
    ("1234567890abcde", 0x1B3503FD), ("1234567890abcdef", 0x1F980463), ("1234567890abcefg", 0x1F9E0466),
    ("1234567890abcefgh", 0x246C04CE), ("1234567890abcefghi", 0x29A30537),
    ("1234567890abcefghij", 0x2F4405A1), ("1234567890abcefghijk", 0x3550060C),
    ("1234567890abcefghijkl", 0x3BC80678), ("1234567890abcefghijklm", 0x42AD06E5),
    ("1234567890abcefghijklmn", 0x4A000753),
    ("1234567890abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", 0x8C40150C),
    ("1234567890", 187433486), ("121231234", 147325380),
]

KAT = {
    "inflate": [{"in": a, "out": b, "zlib": z} for a, b, z in INFLATE],
    "inflate_dict_abcdefgh": [{"in": a, "out": b, "use_dict": d} for a, b, d in INFLATE_DICT],
    "adler32": [{"str": s, "adler": v} for s, v in ADLER],
    "adler32_files": {"adler32Test.txt": 0x5D9BAF5D, "adler32Test2.txt": 0xD6A07E29,
                      "dictionary32768.txt": 4072834167},
    # Test.lua:2210-2245: input = bytes 0..255 then 255..0
    "sizes_updown": {"deflate_default": 517, "deflate_fixed": 542, "zlib_fixed": 548,
                     "deflate_dynamic": 536, "zlib_dynamic": 542},
    # Test.lua:2223-2228: input = "a"*1000
    "sizes_a1000": {"deflate_default": 10, "deflate_huffman_only": 138, "zlib_default": 16,
                    "zlib_huffman_only": 144},
    # Test.lua:2694-2707 upper bounds, levels 0..9, smalltest_no_newline.txt
    "smalltest_bounds": [28458, 7467, 7011, 6740, 6401, 5992, 5884, 5829, 5820, 5820],
    # docs/benchmark.md:177-180 / 243-246 compress ratios, levels 1/5/8
    "ratios": {"encode_6bit_weakaura.txt": [3.15, 3.68, 3.71], "totalrp3.txt": [6.31, 7.64, 8.14]},
}

def main():
    for src, dst in COPY:
        shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(KAT, f, indent=0)
    print("wrote", len(COPY), "files + kat.json")

if __name__ == "__main__":
    main()
