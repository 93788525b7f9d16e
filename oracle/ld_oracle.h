/*
 * ld_oracle.h -- CPU restatement of LibDeflate's compress / decompress hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (libdeflate_amd/,
 * libdeflate_hip.so) includes, links or calls this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / reported baseline.
 *
 * Every function cites the reference lines it restates (LD = LibDeflate.lua in
 * the upstream repository, v1.0.2-release).
 *
 * Parity pinning: see oracle/README.md -- the restatement reproduces the four
 * LibDeflate-made golden streams of tests/data/reference/, the nine exact sizes
 * of tests/Test.lua:2210-2245, the ten bounds of :2694-2707, the Adler-32 KATs of
 * :1707-1739, the 6-bit golden pair of :2078-2083 and every inflate vector of
 * :1311-1634 (tests/test_oracle_golden.py).
 */
#ifndef LD_ORACLE_H
#define LD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* strategy values (LD:704-705) */
#define LDO_STRATEGY_DEFAULT 0
#define LDO_STRATEGY_FIXED 1
#define LDO_STRATEGY_HUFFMAN_ONLY 2
#define LDO_STRATEGY_DYNAMIC 3

/* LD:401-442 */
uint32_t ldo_adler32(const uint8_t *s, size_t n);

/* Upper bound of the compressed size for n input bytes, any level/strategy
 * (forced "fixed"/"dynamic" can exceed the stored size, LD:1874-1894). */
size_t ldo_compress_bound(size_t n);

/* LD:1960-2024.  zlib=0 raw deflate, 1 zlib container.  level -1 = default by
 * size (LD:1782-1790).  dict may be NULL.  Returns 0, or -1 if cap too small. */
int ldo_compress(const uint8_t *in, size_t n, int zlib, int level, int strategy,
                 const uint8_t *dict, size_t dict_len, uint8_t *out, size_t cap,
                 size_t *out_len, int *pad_bits);

/* LD:2648-2727.  Returns the LibDeflate status: 0 ok (out_len, bytes_left set),
 * 2 / -1..-17 as in SURVEY Q11, or 1 if cap is too small (puff's code for output
 * space exhausted; out_len then holds the bytes that would have been needed so
 * far -- no reference counterpart, Lua strings grow).  out may be NULL with
 * cap 0 for a sizing pass (status 1 is then not raised). */
int ldo_decompress(const uint8_t *in, size_t n, int zlib, const uint8_t *dict,
                   size_t dict_len, uint8_t *out, size_t cap, size_t *out_len,
                   size_t *bytes_left);

/* LD:3311-3353.  Returns number of chars written (out needs ceil(4n/3)+1). */
size_t ldo_print6_encode(const uint8_t *in, size_t n, uint8_t *out);
/* LD:3365-3417.  Returns 0 and *out_len, or -1 for nil. */
int ldo_print6_decode(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len);

/* Debug/inspection: tokenise one whole stream (all blocks) and return the
 * (length,distance) token list; literal = (0, byte).  Used by tests to localise
 * mismatches (LZ77 vs Huffman/bitpack).  Returns token count or -1. */
long ldo_tokens(const uint8_t *in, size_t n, int level, const uint8_t *dict,
                size_t dict_len, uint16_t *tok_len, uint16_t *tok_dist, long cap);

/* ---- CreateCodec (LD:2921-3056) and the two WoW codecs built on it (LD:3058-3158) ----
 * Test infrastructure like the rest of this file.  The codec is the set of tables the Lua closure
 * holds: encode_translate (one or two bytes per input byte) and, for Decode, the fail set
 * (reserved_chars) and the ordered list of gsub passes: an optional single-byte translate, then one
 * "escape byte + suffix" pass per escape character that was used. */
typedef struct {
  uint8_t enc_len[256];      /* 1: byte (possibly mapped) -> enc0; 2: escape enc0 + suffix enc1 */
  uint8_t enc0[256], enc1[256];
  uint8_t reserved[256];     /* decode_fail_pattern */
  int has_map;               /* first decode pass: single-byte translate */
  uint8_t map_back[256];     /* identity for bytes that are not mapped */
  int npass;                 /* escape passes */
  uint8_t pass_escape[256];
  uint16_t pass_dec[256][256]; /* suffix -> original byte, 0xFFFF = not in the set */
} ldo_codec;

/* Returns 0 and fills *c, or -1 with *err = the reference's message (second return of CreateCodec). */
int ldo_codec_build(const uint8_t *reserved, size_t nr, const uint8_t *escape, size_t ne,
                    const uint8_t *map, size_t nm, ldo_codec *c, const char **err);
/* codec:Encode; out needs 2n bytes.  Returns the encoded length. */
size_t ldo_codec_encode(const ldo_codec *c, const uint8_t *in, size_t n, uint8_t *out);
/* codec:Decode; out needs n bytes.  Returns 0 and *out_len, or -1 for nil. */
int ldo_codec_decode(const ldo_codec *c, const uint8_t *in, size_t n, uint8_t *out, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
