"""Command line of the reference (LibDeflate.lua:3453-3603) over the MI355X engine:

    python -m libdeflate_amd.cli [OPTION] [INPUT] [OUTPUT]

Same options, same messages on stderr, same exit codes; files only (the reference has no stdin/stdout mode either)."""
import sys

VERSION_TEXT = ("libdeflate_amd: MI355X engine with the interface and the output of LibDeflate 1.0.2-release\n"
                "(LibDeflate is Copyright (C) 2018-2020 Haoqian He, zlib License)")
HELP_TEXT = (VERSION_TEXT + "\nUsage: python -m libdeflate_amd.cli [OPTION] [INPUT] [OUTPUT]\n"
             "  -0    store only. no compression.\n"
             "  -1    fastest compression.\n"
             "  -9    slowest and best compression.\n"
             "  -d    do decompression instead of compression.\n"
             "  --dict <filename> specify the file that contains the entire preset dictionary.\n"
             "  -h    give this help.\n"
             "  --strategy <fixed/huffman_only/dynamic> specify a special compression strategy.\n"
             "  -v    print the version and copyright info.\n"
             "  --zlib  use zlib format instead of raw deflate.\n")


def main(argv, stdout=sys.stdout, stderr=sys.stderr):
    """Returns the exit code.  Mirrors the argument loop of LibDeflate.lua:3470-3546."""
    inp = out_name = None
    in_data = None
    is_zlib = is_decompress = False
    level = strategy = None
    dict_bytes = None
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-h":
            stdout.write(HELP_TEXT + "\n")
            return 0
        elif a == "-v":
            stdout.write(VERSION_TEXT + "\n")
            return 0
        elif len(a) == 2 and a[0] == "-" and a[1].isdigit():
            level = int(a[1])
        elif a == "-d":
            is_decompress = True
        elif a == "--dict":
            i += 1
            if i >= len(argv):
                stderr.write("You must speicify the dict filename")  # (sic, LibDeflate.lua:3498)
                return 1
            try:
                with open(argv[i], "rb") as f:
                    dict_bytes = f.read()
            except OSError as e:
                stderr.write("LibDeflate: Cannot read the dictionary file '%s': %s" % (argv[i], e.strerror))
                return 1
        elif a == "--strategy":
            i += 1
            strategy = argv[i] if i < len(argv) else None
        elif a == "--zlib":
            is_zlib = True
        elif a.startswith("-"):
            stderr.write("LibDeflate: Invalid argument: %s" % a)
            return 1
        elif inp is None:
            try:
                with open(a, "rb") as f:
                    in_data = f.read()
                inp = a
            except OSError as e:
                stderr.write("LibDeflate: Cannot read the file '%s': %s" % (a, e.strerror))
                return 1
        elif out_name is None:
            try:
                open(a, "wb").close()
                out_name = a
            except OSError as e:
                stderr.write("LibDeflate: Cannot write the file '%s': %s" % (a, e.strerror))
                return 1
        i += 1
    if inp is None or out_name is None:
        stderr.write("LibDeflate: You must specify both input and output files.")
        return 1

    from .LibDeflate import LibDeflate as LD
    dictionary = None
    if dict_bytes is not None:  # LibDeflate.lua:3514-3515
        dictionary = LD.CreateDictionary(dict_bytes, len(dict_bytes), LD.Adler32(dict_bytes))
    configs = {}
    if level is not None:
        configs["level"] = level
    if strategy is not None:
        configs["strategy"] = strategy
    if not is_decompress:
        if not is_zlib:
            res = LD.CompressDeflate(in_data, configs) if dictionary is None else \
                LD.CompressDeflateWithDict(in_data, dictionary, configs)
        else:
            res = LD.CompressZlib(in_data, configs) if dictionary is None else \
                LD.CompressZlibWithDict(in_data, dictionary, configs)
    else:
        if not is_zlib:
            res = LD.DecompressDeflate(in_data) if dictionary is None else \
                LD.DecompressDeflateWithDict(in_data, dictionary)
        else:
            res = LD.DecompressZlib(in_data) if dictionary is None else \
                LD.DecompressZlibWithDict(in_data, dictionary)
    data = res[0]
    if data is None:
        stderr.write("LibDeflate: Decompress fails.")
        return 1
    with open(out_name, "wb") as f:
        f.write(data)
    stderr.write("Successfully writes %d bytes" % len(data))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
