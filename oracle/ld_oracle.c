/*
 * ld_oracle.c -- CPU restatement (plain C, integer only) of LibDeflate's
 * compress / decompress / Adler-32 / 6-bit print codec.
 *
 * TEST INFRASTRUCTURE ONLY (see ld_oracle.h).  This is the checker the HIP
 * kernels are compared against; it is never the thing measured or shipped.
 *
 * "LD:a-b" = lines a..b of LibDeflate.lua (upstream v1.0.2-release).
 * The Lua source works on 1-based positions and keeps the input in a re-based
 * table (string_table[index - offset]); this restatement keeps absolute 1-based
 * positions so that W(p) below is string_table[p - offset].
 */
#include "ld_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables */

/* LD:657-669 {use_lazy, good_prev_length, max_lazy|max_insert, nice, max_chain} */
typedef struct {
  int lazy, good, max_lazy, nice, max_chain;
} LevelCfg;
static const LevelCfg LEVELS[10] = {
    {0, 0, 0, 0, 0},      {0, 0, 4, 8, 4},       {0, 0, 5, 18, 8},
    {0, 0, 6, 32, 32},    {1, 4, 4, 16, 16},     {1, 8, 16, 32, 32},
    {1, 8, 16, 128, 128}, {1, 8, 32, 128, 256},  {1, 32, 128, 258, 1024},
    {1, 32, 258, 258, 4096}};

/* LD:188-200 */
static const int LEN_BASE[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,
                                 15, 17, 19, 23, 27, 31, 35, 43, 51,  59,
                                 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const int LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                  2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
/* LD:203-269 */
static const int DIST_BASE[30] = {1,    2,    3,    4,    5,    7,     9,    13,
                                  17,   25,   33,   49,   65,   97,    129,  193,
                                  257,  385,  513,  769,  1025, 1537,  2049, 3073,
                                  4097, 6145, 8193, 12289, 16385, 24577};
static const int DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,
                                   4, 4, 5, 5, 6, 6, 7,  7,  8,  8,
                                   9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
/* LD:273-275 */
static const int CL_ORDER[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                 11, 4,  12, 3, 13, 2, 14, 1, 15};

/* LD:341-366: length 3..258 -> code / extra bits (258 -> 285, no extra). */
static void length_code(int len, int *code, int *ebits, int *eval) {
  int c;
  if (len == 258) {
    *code = 285;
    *ebits = 0;
    *eval = 0;
    return;
  }
  c = 0;
  while (c + 1 < 28 && LEN_BASE[c + 1] <= len) c++;
  *code = 257 + c;
  *ebits = LEN_EXTRA[c];
  *eval = len - LEN_BASE[c];
}
/* LD:370-393, LD:1408-1433 */
static void dist_code(int dist, int *code, int *ebits, int *eval) {
  int c = 0;
  while (c + 1 < 30 && DIST_BASE[c + 1] <= dist) c++;
  *code = c;
  *ebits = DIST_EXTRA[c];
  *eval = dist - DIST_BASE[c];
}

/* ------------------------------------------------------------------ adler */

/* LD:401-442 (the unrolled form computes the same two sums mod 65521). */
uint32_t ldo_adler32(const uint8_t *s, size_t n) {
  uint32_t a = 1, b = 0;
  size_t i = 0;
  while (i < n) {
    size_t chunk = n - i;
    size_t k;
    if (chunk > 5552) chunk = 5552;
    for (k = 0; k < chunk; k++) {
      a += s[i + k];
      b += a;
    }
    a %= 65521u;
    b %= 65521u;
    i += chunk;
  }
  return (b << 16) | a;
}

/* ------------------------------------------------------------------ writer */

/* LD:736-832 CreateWriter: LSB-first packing, 1-bit padding. */
typedef struct {
  uint8_t *out;
  size_t cap, len;
  uint64_t cache;
  int cache_bitlen;
  uint64_t total_bitlen;
  int overflow;
} Writer;

static void w_byte(Writer *w, unsigned b) {
  if (w->len < w->cap)
    w->out[w->len] = (uint8_t)b;
  else
    w->overflow = 1;
  w->len++;
}
/* LD:749-766 */
static void write_bits(Writer *w, uint32_t value, int bitlen) {
  w->cache |= (uint64_t)value << w->cache_bitlen;
  w->cache_bitlen += bitlen;
  w->total_bitlen += (uint64_t)bitlen;
  while (w->cache_bitlen >= 8) {
    w_byte(w, (unsigned)(w->cache & 0xFF));
    w->cache >>= 8;
    w->cache_bitlen -= 8;
  }
}
/* LD:771-781 (only called on a byte boundary) */
static void write_string(Writer *w, const uint8_t *s, size_t n) {
  size_t i;
  for (i = 0; i < n; i++) w_byte(w, s[i]);
  w->total_bitlen += 8u * (uint64_t)n;
}
/* LD:793-816 pad the last partial byte with 1-bits; returns pad count.
 * count_padding: BYTE_BOUNDARY mode adds the padding to total_bitlen. */
static int flush_pad(Writer *w, int count_padding) {
  int pad = (8 - w->cache_bitlen % 8) % 8;
  if (w->cache_bitlen > 0) {
    w->cache |= ((1ull << pad) - 1ull) << w->cache_bitlen;
    w_byte(w, (unsigned)(w->cache & 0xFF));
    w->cache = 0;
    w->cache_bitlen = 0;
  }
  if (count_padding) w->total_bitlen += (uint64_t)pad;
  return pad;
}

/* ------------------------------------------------------------------ huffman */

#define NONE INT_MIN

typedef struct {
  long w; /* weight, later bit length (LD:978) */
  int sym, left, right;
} HNode;

/* LD:842-856 */
static void heap_push(HNode *nd, int *heap, int e, int heap_size) {
  long value;
  int pos, parent;
  heap_size++;
  heap[heap_size] = e;
  value = nd[e].w;
  pos = heap_size;
  parent = pos / 2;
  while (parent >= 1 && nd[heap[parent]].w > value) {
    int t = heap[parent];
    heap[parent] = e;
    heap[pos] = t;
    pos = parent;
    parent = parent / 2;
  }
}
/* LD:863-903 */
static int heap_pop(HNode *nd, int *heap, int heap_size) {
  int top = heap[1];
  int e = heap[heap_size];
  long value = nd[e].w;
  int pos = 1, l = 2, r = 3;
  heap[1] = e;
  heap[heap_size] = top;
  heap_size--;
  while (l <= heap_size) {
    int lc = heap[l];
    if (r <= heap_size && nd[heap[r]].w < nd[lc].w) {
      int rc = heap[r];
      if (nd[rc].w < value) {
        heap[r] = e;
        heap[pos] = rc;
        pos = r;
        l = pos * 2;
        r = l + 1;
      } else
        break;
    } else {
      if (nd[lc].w < value) {
        heap[l] = e;
        heap[pos] = lc;
        pos = l;
        l = pos * 2;
        r = l + 1;
      } else
        break;
    }
  }
  return top;
}

/* LD:913-948: canonical codes, bit-reversed.  bitlens[sym]==0 means absent. */
static void codes_from_bitlens(const int *bl_count, const int *bitlens, int max_symbol,
                               int max_bitlen, int *codes) {
  int next_code[17];
  int code = 0, b, s;
  for (b = 1; b <= max_bitlen; b++) {
    code = (code + bl_count[b - 1]) * 2;
    next_code[b] = code;
  }
  for (s = 0; s <= max_symbol; s++) {
    int len = bitlens[s];
    if (len) {
      int c = next_code[len]++, rev = 0, k;
      for (k = 0; k < len; k++) {
        rev = (rev << 1) | (c & 1);
        c >>= 1;
      }
      codes[s] = rev;
    }
  }
}

/* LD:968-1085 GetHuffmanBitlenAndCode.  counts[0..max_symbol]; outputs bitlens
 * (0 = no code), codes; returns max_non_zero_bitlen_symbol (-1 if none). */
static int huffman_bitlen_and_code(const long *counts, int max_bitlen, int max_symbol,
                                   int *bitlens, int *codes) {
  HNode nd[2 * 288];
  int leafs[288], heap[288 + 2], fifo[2 * 288];
  int bl_count[17];
  int n = 0, s, i, heap_size, nnodes, max_nz = -1;
  int fifo_size, index, overflow = 0;
  for (s = 0; s <= max_symbol; s++) {
    bitlens[s] = 0;
    codes[s] = 0;
  }
  for (s = 0; s <= max_symbol; s++)
    if (counts[s] > 0) {
      nd[n].w = counts[s];
      nd[n].sym = s;
      nd[n].left = nd[n].right = -1;
      n++;
    }
  if (n == 0) return -1; /* LD:989-991 */
  if (n == 1) {          /* LD:992-1000 */
    bitlens[nd[0].sym] = 1;
    codes[nd[0].sym] = 0;
    return nd[0].sym;
  }
  /* LD:1002 sort by (count, symbol): insertion sort, symbols already ascending
   * so a stable sort on count reproduces SortByFirstThenSecond. */
  for (i = 0; i < n; i++) leafs[i] = i;
  for (i = 1; i < n; i++) {
    int e = leafs[i], j = i - 1;
    while (j >= 0 && nd[leafs[j]].w > nd[e].w) {
      leafs[j + 1] = leafs[j];
      j--;
    }
    leafs[j + 1] = e;
  }
  heap_size = n;
  for (i = 1; i <= n; i++) heap[i] = leafs[i - 1];
  nnodes = n;
  while (heap_size > 1) { /* LD:1006-1015 */
    int l = heap_pop(nd, heap, heap_size), r;
    heap_size--;
    r = heap_pop(nd, heap, heap_size);
    heap_size--;
    nd[nnodes].w = nd[l].w + nd[r].w;
    nd[nnodes].sym = -1;
    nd[nnodes].left = l;
    nd[nnodes].right = r;
    heap_push(nd, heap, nnodes, heap_size);
    nnodes++;
    heap_size++;
  }
  /* LD:1021-1053 breadth-first depth assignment */
  for (i = 0; i <= 16; i++) bl_count[i] = 0;
  fifo[0] = heap[1];
  fifo_size = 1;
  index = 0;
  nd[heap[1]].w = 0;
  while (index < fifo_size) {
    int e = fifo[index];
    long bitlen = nd[e].w;
    if (nd[e].left >= 0) {
      fifo[fifo_size++] = nd[e].left;
      nd[nd[e].left].w = bitlen + 1;
    }
    if (nd[e].right >= 0) {
      fifo[fifo_size++] = nd[e].right;
      nd[nd[e].right].w = bitlen + 1;
    }
    index++;
    if (bitlen > max_bitlen) {
      overflow++;
      bitlen = max_bitlen;
    }
    if (nd[e].sym >= 0) {
      bitlens[nd[e].sym] = (int)bitlen;
      if (nd[e].sym > max_nz) max_nz = nd[e].sym;
      bl_count[bitlen]++;
    }
  }
  /* LD:1057-1079 zlib gen_bitlen style fix-up */
  if (overflow > 0) {
    int bitlen;
    do {
      bitlen = max_bitlen - 1;
      while (bitlen > 0 && bl_count[bitlen] == 0) bitlen--;
      if (bitlen <= 0) break; /* the Lua loop would not terminate here */
      bl_count[bitlen]--;
      bl_count[bitlen + 1] += 2;
      bl_count[max_bitlen]--;
      overflow -= 2;
    } while (overflow > 0);
    index = 0;
    for (bitlen = max_bitlen; bitlen >= 1; bitlen--) {
      int k = bl_count[bitlen];
      while (k > 0) {
        bitlens[nd[leafs[index]].sym] = bitlen;
        k--;
        index++;
      }
    }
  }
  codes_from_bitlens(bl_count, bitlens, max_symbol, max_bitlen, codes);
  return max_nz;
}

/* LD:1099-1174 RunLengthEncodeHuffmanBitlen.  Returns number of rle codes. */
static int rle_encode_bitlens(const int *lbl, int max_l, const int *dbl, int max_d,
                              int *rle_codes, int *rle_extra, long *rle_counts) {
  int nrle = 0, nextra = 0, code, prev = -1 /* nil */, count = 0, max_code, i;
  for (i = 0; i < 19; i++) rle_counts[i] = 0;
  if (max_d < 0) max_d = 0;
  max_code = max_l + max_d + 1;
  for (code = 0; code <= max_code + 1; code++) {
    /* len == -2 stands for Lua nil at code == max_code+1 (prev==-1 is the
     * initial nil; the two never meet because max_code+1 >= 2). */
    int len = (code <= max_l) ? lbl[code]
                              : ((code <= max_code) ? dbl[code - max_l - 1] : -2);
    if (len == prev) {
      count++;
      if (len != 0 && count == 6) {
        rle_codes[nrle++] = 16;
        rle_extra[nextra++] = 3;
        rle_counts[16]++;
        count = 0;
      } else if (len == 0 && count == 138) {
        rle_codes[nrle++] = 18;
        rle_extra[nextra++] = 127;
        rle_counts[18]++;
        count = 0;
      }
    } else {
      if (count == 1) {
        rle_codes[nrle++] = prev;
        rle_counts[prev]++;
      } else if (count == 2) {
        rle_codes[nrle++] = prev;
        rle_codes[nrle++] = prev;
        rle_counts[prev] += 2;
      } else if (count >= 3) {
        int rc = (prev != 0) ? 16 : (count <= 10 ? 17 : 18);
        rle_codes[nrle++] = rc;
        rle_counts[rc]++;
        rle_extra[nextra++] = (count <= 10) ? (count - 3) : (count - 11);
      }
      prev = len;
      if (len > 0) { /* "len and len ~= 0" */
        rle_codes[nrle++] = len;
        rle_counts[len]++;
        count = 0;
      } else {
        count = 1;
      }
    }
  }
  return nrle;
}

/* ------------------------------------------------------------------ chains */

/* Exact-trigram chains (Q1): an open-addressing map trigram -> newest inserted
 * position, plus a prev-link per position.  Equivalent to the Lua arrays
 * hash_tables[hash] = {pos, ...} walked from the back (LD:1329-1398). */
typedef struct {
  int32_t *keys;  /* -1 = empty */
  int32_t *heads; /* NONE = chain exists but is empty */
  uint32_t mask;
  int shift;
} TriMap;

static int trimap_init(TriMap *m, size_t expected) {
  uint32_t sz = 1024;
  int bits = 10;
  size_t i;
  while ((size_t)sz < expected * 2 + 16) {
    sz <<= 1;
    bits++;
  }
  m->keys = (int32_t *)malloc(sizeof(int32_t) * sz);
  m->heads = (int32_t *)malloc(sizeof(int32_t) * sz);
  if (!m->keys || !m->heads) return -1;
  for (i = 0; i < sz; i++) m->keys[i] = -1;
  m->mask = sz - 1;
  m->shift = 32 - bits;
  return 0;
}
static void trimap_free(TriMap *m) {
  free(m->keys);
  free(m->heads);
  m->keys = m->heads = NULL;
}
static int32_t *trimap_slot(TriMap *m, uint32_t key, int create) {
  uint32_t h = (key * 2654435761u) >> m->shift;
  for (;;) {
    if (m->keys[h] == (int32_t)key) return &m->heads[h];
    if (m->keys[h] == -1) {
      if (!create) return NULL;
      m->keys[h] = (int32_t)key;
      m->heads[h] = NONE;
      return &m->heads[h];
    }
    h = (h + 1) & m->mask;
  }
}

/* ------------------------------------------------------------------ LZ77 */

typedef struct {
  uint16_t lcode;  /* 0-255 literal, 256 EOB, 257-285 length code */
  uint16_t lextra; /* extra-bit value of the length code */
  uint8_t dcode;
  uint16_t dextra;
  uint16_t len, dist; /* kept for ldo_tokens */
} Token;

typedef struct {
  const uint8_t *in;
  long n;
  const uint8_t *dict; /* NULL once the first block is done (LD:1917) */
  long dlen;
  TriMap stream_map;
  int32_t *stream_prev; /* index pos+1, pos in [-1, n] */
  TriMap dict_map;      /* LD:531-595 */
  int32_t *dict_prev;   /* index pos+dlen, pos in [-(dlen-1), -2] */
  int have_dict_map;
} LzState;

/* string_table[p - offset] (LD:1184-1197, 1825-1831): p>=1 input, p<=0 dict tail */
static int W(const LzState *st, long p) {
  if (p >= 1) return (p <= st->n) ? st->in[p - 1] : -1; /* -1 == nil */
  if (st->dict && p >= 1 - st->dlen && p >= -257) return st->dict[st->dlen + p - 1];
  return -1;
}
static uint32_t W0(const LzState *st, long p) {
  int v = W(st, p);
  return v < 0 ? 0u : (uint32_t)v;
}
static void stream_insert(LzState *st, uint32_t hash, long pos) {
  int32_t *slot = trimap_slot(&st->stream_map, hash, 1);
  st->stream_prev[pos + 1] = *slot;
  *slot = (int32_t)pos;
}

/* LD:1241-1484 GetBlockLZ77Result.  Appends tokens (incl. EOB) to tok[];
 * fills lcounts[286], dcounts[30].  Returns token count. */
static long lz77_block(LzState *st, int level, long block_start, long block_end, Token *tok,
                       long *lcounts, long *dcounts) {
  const LevelCfg *cfg = &LEVELS[level];
  const int lazy = cfg->lazy;
  const long max_insert = lazy ? 2147483646L : cfg->max_lazy; /* LD:1249-1250 */
  /* LD:1251-1252: max_chain - ((max_chain % 4) / 4); Lua numbers are doubles */
  const double good_chain = (double)cfg->max_chain - (double)(cfg->max_chain % 4) / 4.0;
  uint32_t hash;
  long ntok = 0, index, index_end;
  int match_available = 0;
  long prev_len = 0, prev_dist = 0, cur_len = 0, cur_dist = 0;
  const uint8_t *dict = st->dict;
  const long dlen = dict ? st->dlen : 0;
  int i;
  for (i = 0; i < 286; i++) lcounts[i] = 0;
  for (i = 0; i < 30; i++) dcounts[i] = 0;

  if (dict) { /* LD:1260-1286 seed the two boundary trigrams */
    if (block_end >= block_start && dlen >= 2) {
      hash = (uint32_t)dict[dlen - 2] * 65536u + (uint32_t)dict[dlen - 1] * 256u + W0(st, 1);
      stream_insert(st, hash, -1);
    }
    if (block_end >= block_start + 1 && dlen >= 1) {
      hash = (uint32_t)dict[dlen - 1] * 65536u + W0(st, 1) * 256u + W0(st, 2);
      stream_insert(st, hash, 0);
    }
  }

  hash = W0(st, block_start) * 256u + W0(st, block_start + 1); /* LD:1290 */
  index = block_start;
  index_end = block_end + (lazy ? 1 : 0);

  while (index <= index_end) { /* LD:1318 */
    int32_t *slot;
    long chain_head;     /* newest candidate or NONE (chain_index == 0) */
    int in_dict_chain = 0;
    prev_len = cur_len;
    prev_dist = cur_dist;
    cur_len = 0;
    hash = (hash * 256u + W0(st, index + 2)) & 0xFFFFFFu; /* LD:1325 */

    slot = trimap_slot(&st->stream_map, hash, 0);
    if (!slot) { /* LD:1331-1340 */
      slot = trimap_slot(&st->stream_map, hash, 1);
      chain_head = NONE;
      if (dict && st->have_dict_map) {
        int32_t *ds = trimap_slot(&st->dict_map, hash, 0);
        if (ds && *ds != NONE) {
          chain_head = *ds;
          in_dict_chain = 1;
        }
      }
    } else { /* LD:1342-1344 */
      chain_head = *slot;
    }
    if (index <= block_end) { /* LD:1347 */
      st->stream_prev[index + 1] = *slot;
      *slot = (int32_t)index;
    }

    if (chain_head != NONE && index + 2 <= block_end &&
        (!lazy || prev_len < cfg->max_lazy)) { /* LD:1349-1350 */
      double depth = (lazy && prev_len >= cfg->good) ? good_chain : (double)cfg->max_chain;
      long max_len_minus_one = block_end - index;
      long p = chain_head;
      if (max_len_minus_one >= 257) max_len_minus_one = 257;
      max_len_minus_one += index; /* last comparable position */
      while (p != NONE && depth > 0) { /* LD:1362 */
        long prev = p;
        if (index - prev > 32768) break; /* LD:1365 */
        if (prev < index) {
          long sj = index + 3, j;
          if (prev >= -257) { /* LD:1369-1375 */
            long pj = prev + 3;
            while (sj <= max_len_minus_one && W(st, pj) == W(st, sj)) {
              sj++;
              pj++;
            }
          } else { /* LD:1376-1383 compare inside dict_string_table */
            long pj = dlen + 3 + prev; /* 1-based index into the dictionary */
            while (sj <= max_len_minus_one && pj <= dlen && (int)dict[pj - 1] == W(st, sj)) {
              sj++;
              pj++;
            }
          }
          j = sj - index;
          if (j > cur_len) {
            cur_len = j;
            cur_dist = index - prev;
          }
          if (cur_len >= cfg->nice) break; /* LD:1389 */
        }
        /* LD:1392-1397 */
        p = in_dict_chain ? st->dict_prev[prev + dlen] : st->stream_prev[prev + 1];
        depth -= 1;
        if (p == NONE && prev > 0 && dict && st->have_dict_map && !in_dict_chain) {
          int32_t *ds = trimap_slot(&st->dict_map, hash, 0);
          in_dict_chain = 1;
          p = (ds && *ds != NONE) ? *ds : NONE;
        }
      }
    }

    if (!lazy) { /* LD:1401 */
      prev_len = cur_len;
      prev_dist = cur_dist;
    }
    if ((!lazy || match_available) &&
        (prev_len > 3 || (prev_len == 3 && prev_dist < 4096)) && cur_len <= prev_len) {
      /* LD:1405-1464 emit match */
      int code, lbits, lval, dcode, dbits, dval;
      long i2;
      Token *t = &tok[ntok++];
      length_code((int)prev_len, &code, &lbits, &lval);
      dist_code((int)prev_dist, &dcode, &dbits, &dval);
      t->lcode = (uint16_t)code;
      t->lextra = (uint16_t)lval;
      t->dcode = (uint8_t)dcode;
      t->dextra = (uint16_t)dval;
      t->len = (uint16_t)prev_len;
      t->dist = (uint16_t)(prev_dist & 0xFFFF); /* 32768 fits */
      lcounts[code]++;
      dcounts[dcode]++;
      for (i2 = index + 1; i2 <= index + prev_len - (lazy ? 2 : 1); i2++) {
        hash = (hash * 256u + W0(st, i2 + 2)) & 0xFFFFFFu;
        if (prev_len <= max_insert) stream_insert(st, hash, i2);
      }
      index = index + prev_len - (lazy ? 1 : 0);
      match_available = 0;
    } else if (!lazy || match_available) { /* LD:1465-1471 */
      int code = W(st, lazy ? index - 1 : index);
      Token *t = &tok[ntok++];
      t->lcode = (uint16_t)code;
      t->lextra = 0;
      t->dcode = 0;
      t->dextra = 0;
      t->len = 0;
      t->dist = (uint16_t)code;
      lcounts[code]++;
      index++;
    } else { /* LD:1472-1475 */
      match_available = 1;
      index++;
    }
  }
  { /* LD:1478-1481 */
    Token *t = &tok[ntok++];
    memset(t, 0, sizeof(*t));
    t->lcode = 256;
    lcounts[256]++;
  }
  return ntok;
}

/* ------------------------------------------------------------------ blocks */

typedef struct {
  int HLIT, HDIST, HCLEN;
  int cl_bitlens[19], cl_codes[19];
  int rle_codes[320], rle_extra[320], nrle;
  int l_bitlens[286], l_codes[286];
  int d_bitlens[30], d_codes[30];
} DynHeader;

/* LD:1491-1521 */
static void dynamic_header(const long *lcounts, const long *dcounts, DynHeader *h) {
  long rle_counts[19];
  int max_l, max_d, i, hclen = 0;
  max_l = huffman_bitlen_and_code(lcounts, 15, 285, h->l_bitlens, h->l_codes);
  max_d = huffman_bitlen_and_code(dcounts, 15, 29, h->d_bitlens, h->d_codes);
  h->nrle =
      rle_encode_bitlens(h->l_bitlens, max_l, h->d_bitlens, max_d, h->rle_codes, h->rle_extra, rle_counts);
  huffman_bitlen_and_code(rle_counts, 7, 18, h->cl_bitlens, h->cl_codes);
  for (i = 1; i <= 19; i++)
    if (h->cl_bitlens[CL_ORDER[i - 1]] != 0) hclen = i;
  h->HCLEN = hclen - 4;
  h->HLIT = max_l + 1 - 257;
  h->HDIST = max_d + 1 - 1;
  if (h->HDIST < 0) h->HDIST = 0;
}

static int rle_extra_bitlen(int code) { return code == 16 ? 2 : (code == 17 ? 3 : 7); }

/* LD:1526-1567 */
static uint64_t dynamic_block_size(const Token *tok, long ntok, const DynHeader *h) {
  uint64_t bits = 17 + (uint64_t)(h->HCLEN + 4) * 3;
  long i;
  for (i = 0; i < h->nrle; i++) {
    int c = h->rle_codes[i];
    bits += (uint64_t)h->cl_bitlens[c];
    if (c >= 16) bits += (uint64_t)rle_extra_bitlen(c);
  }
  for (i = 0; i < ntok; i++) {
    int c = tok[i].lcode;
    bits += (uint64_t)h->l_bitlens[c];
    if (c > 256) {
      int dc = tok[i].dcode;
      if (c > 264 && c < 285) bits += (uint64_t)LEN_EXTRA[c - 257];
      bits += (uint64_t)h->d_bitlens[dc];
      if (dc > 3) bits += (uint64_t)(dc / 2 - 1);
    }
  }
  return bits;
}
/* LD:2826-2849 fixed code lengths / codes */
static int fix_l_bitlen(int s) { return s <= 143 ? 8 : (s <= 255 ? 9 : (s <= 279 ? 7 : 8)); }
static int FIX_L_CODE[288], FIX_D_CODE[32], fix_ready = 0;
static void fix_init(void) {
  int bl[288], cnt[17], i;
  if (fix_ready) return;
  for (i = 0; i <= 16; i++) cnt[i] = 0;
  for (i = 0; i < 288; i++) {
    bl[i] = fix_l_bitlen(i);
    cnt[bl[i]]++;
  }
  codes_from_bitlens(cnt, bl, 287, 9, FIX_L_CODE);
  for (i = 0; i <= 16; i++) cnt[i] = 0;
  for (i = 0; i < 32; i++) bl[i] = 5;
  cnt[5] = 32;
  codes_from_bitlens(cnt, bl, 31, 5, FIX_D_CODE);
  fix_ready = 1;
}
/* LD:1646-1670 */
static uint64_t fixed_block_size(const Token *tok, long ntok) {
  uint64_t bits = 3;
  long i;
  for (i = 0; i < ntok; i++) {
    int c = tok[i].lcode;
    bits += (uint64_t)fix_l_bitlen(c);
    if (c > 256) {
      int dc = tok[i].dcode;
      if (c > 264 && c < 285) bits += (uint64_t)LEN_EXTRA[c - 257];
      bits += 5;
      if (dc > 3) bits += (uint64_t)(dc / 2 - 1);
    }
  }
  return bits;
}
/* LD:1718-1727 */
static uint64_t store_block_size(long block_start, long block_end, uint64_t total_bitlen) {
  uint64_t t = total_bitlen + 3;
  uint64_t pad = (8 - t % 8) % 8;
  return 3 + pad + 32 + (uint64_t)(block_end - block_start + 1) * 8;
}

/* LD:1571-1640 */
static void write_dynamic_block(Writer *w, int last, const Token *tok, long ntok, const DynHeader *h) {
  long i;
  int xi = 0;
  write_bits(w, last ? 1 : 0, 1);
  write_bits(w, 2, 2);
  write_bits(w, (uint32_t)h->HLIT, 5);
  write_bits(w, (uint32_t)h->HDIST, 5);
  write_bits(w, (uint32_t)h->HCLEN, 4);
  for (i = 1; i <= h->HCLEN + 4; i++) write_bits(w, (uint32_t)h->cl_bitlens[CL_ORDER[i - 1]], 3);
  for (i = 0; i < h->nrle; i++) {
    int c = h->rle_codes[i];
    write_bits(w, (uint32_t)h->cl_codes[c], h->cl_bitlens[c]);
    if (c >= 16) write_bits(w, (uint32_t)h->rle_extra[xi++], rle_extra_bitlen(c));
  }
  for (i = 0; i < ntok; i++) {
    int c = tok[i].lcode;
    write_bits(w, (uint32_t)h->l_codes[c], h->l_bitlens[c]);
    if (c > 256) {
      int dc = tok[i].dcode;
      if (c > 264 && c < 285) write_bits(w, tok[i].lextra, LEN_EXTRA[c - 257]);
      write_bits(w, (uint32_t)h->d_codes[dc], h->d_bitlens[dc]);
      if (dc > 3) write_bits(w, tok[i].dextra, dc / 2 - 1);
    }
  }
}
/* LD:1675-1710 */
static void write_fixed_block(Writer *w, int last, const Token *tok, long ntok) {
  long i;
  write_bits(w, last ? 1 : 0, 1);
  write_bits(w, 1, 2);
  for (i = 0; i < ntok; i++) {
    int c = tok[i].lcode;
    write_bits(w, (uint32_t)FIX_L_CODE[c], fix_l_bitlen(c));
    if (c > 256) {
      int dc = tok[i].dcode;
      if (c > 264 && c < 285) write_bits(w, tok[i].lextra, LEN_EXTRA[c - 257]);
      write_bits(w, (uint32_t)FIX_D_CODE[dc], 5);
      if (dc > 3) write_bits(w, tok[i].dextra, dc / 2 - 1);
    }
  }
}
/* LD:1732-1750 */
static void write_store_block(Writer *w, int last, const uint8_t *in, long block_start, long block_end,
                              uint64_t total_bitlen) {
  uint64_t t = total_bitlen + 3;
  int pad = (int)((8 - t % 8) % 8);
  long size = block_end - block_start + 1;
  write_bits(w, last ? 1 : 0, 1);
  write_bits(w, 0, 2);
  if (pad > 0) write_bits(w, (1u << pad) - 1u, pad);
  write_bits(w, (uint32_t)size, 16);
  write_bits(w, (uint32_t)((255 - size % 256) + (255 - size / 256) * 256), 16);
  write_string(w, in + (block_start - 1), (size_t)size);
}

/* LD:1763-1945 Deflate.  tok_sink (optional) receives every token. */
static int deflate_stream(Writer *w, const uint8_t *in, long n, int level, int strategy,
                          const uint8_t *dict, long dlen, uint16_t *sink_len, uint16_t *sink_dist,
                          long sink_cap, long *sink_n) {
  LzState st;
  Token *tok = NULL;
  long block_start = 0, block_end = 0;
  int is_last = 0, first = 1, rc = 0;
  uint64_t total_bitlen = w->total_bitlen; /* LD:1771 */
  long lcounts[286], dcounts[30];
  DynHeader *hdr = NULL;

  fix_init();
  if (level < 0) level = (n < 2048) ? 7 : (n > 65536 ? 3 : 5); /* LD:1782-1790 */

  memset(&st, 0, sizeof(st));
  st.in = in;
  st.n = n;
  st.dict = dict;
  st.dlen = dlen;
  if (level != 0) {
    long i;
    tok = (Token *)malloc(sizeof(Token) * 65540);
    hdr = (DynHeader *)malloc(sizeof(DynHeader));
    st.stream_prev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 4));
    if (!tok || !hdr || !st.stream_prev || trimap_init(&st.stream_map, (size_t)n + 4) != 0) {
      rc = -2;
      goto done;
    }
    if (dict && strategy != LDO_STRATEGY_HUFFMAN_ONLY) { /* LD:531-595 dictionary chains */
      st.dict_prev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(dlen + 4));
      if (!st.dict_prev || trimap_init(&st.dict_map, (size_t)dlen + 4) != 0) {
        rc = -2;
        goto done;
      }
      st.have_dict_map = 1;
      for (i = 1; i <= dlen - 2; i++) {
        uint32_t h = (uint32_t)dict[i - 1] * 65536u + (uint32_t)dict[i] * 256u + (uint32_t)dict[i + 1];
        int32_t *slot = trimap_slot(&st.dict_map, h, 1);
        long pos = i - dlen;
        st.dict_prev[pos + dlen] = *slot;
        *slot = (int32_t)pos;
      }
    }
  }

  while (!is_last) {
    long ntok = 0;
    uint64_t dyn_bits = 0, fix_bits = 0, store_bits, min_bits;
    int have_huff = 0;
    if (first) { /* LD:1793-1801 */
      block_start = 1;
      block_end = 64 * 1024 - 1;
      first = 0;
    } else {
      block_start = block_end + 1;
      block_end = block_end + 32 * 1024;
    }
    if (block_end >= n) { /* LD:1803-1808 */
      block_end = n;
      is_last = 1;
    }
    if (level != 0) {
      if (strategy == LDO_STRATEGY_HUFFMAN_ONLY) { /* LD:1833-1845 */
        long i;
        for (i = 0; i < 286; i++) lcounts[i] = 0;
        for (i = 0; i < 30; i++) dcounts[i] = 0;
        for (i = block_start; i <= block_end; i++) {
          Token *t = &tok[ntok++];
          memset(t, 0, sizeof(*t));
          t->lcode = in[i - 1];
          t->dist = in[i - 1];
          lcounts[in[i - 1]]++;
        }
        memset(&tok[ntok], 0, sizeof(Token));
        tok[ntok++].lcode = 256;
        lcounts[256]++;
      } else {
        ntok = lz77_block(&st, level, block_start, block_end, tok, lcounts, dcounts);
      }
      if (sink_len) {
        long i;
        for (i = 0; i < ntok; i++) {
          if (tok[i].lcode == 256) continue;
          if (*sink_n < sink_cap) {
            sink_len[*sink_n] = tok[i].len;
            sink_dist[*sink_n] = tok[i].dist;
          }
          (*sink_n)++;
        }
      }
      dynamic_header(lcounts, dcounts, hdr);
      dyn_bits = dynamic_block_size(tok, ntok, hdr);
      fix_bits = fixed_block_size(tok, ntok);
      have_huff = 1;
    }
    store_bits = store_block_size(block_start, block_end, total_bitlen);
    min_bits = store_bits; /* LD:1867-1872 */
    if (have_huff && fix_bits < min_bits) min_bits = fix_bits;
    if (have_huff && dyn_bits < min_bits) min_bits = dyn_bits;

    if (level == 0 || (strategy != LDO_STRATEGY_FIXED && strategy != LDO_STRATEGY_DYNAMIC &&
                       store_bits == min_bits)) { /* LD:1874-1879 */
      write_store_block(w, is_last, in, block_start, block_end, total_bitlen);
      total_bitlen += store_bits;
    } else if (strategy != LDO_STRATEGY_DYNAMIC &&
               (strategy == LDO_STRATEGY_FIXED || fix_bits == min_bits)) { /* LD:1880-1884 */
      write_fixed_block(w, is_last, tok, ntok);
      total_bitlen += fix_bits;
    } else if (strategy == LDO_STRATEGY_DYNAMIC || dyn_bits == min_bits) { /* LD:1885-1893 */
      write_dynamic_block(w, is_last, tok, ntok, hdr);
      total_bitlen += dyn_bits;
    }
    if (w->total_bitlen != total_bitlen) { /* LD:1902 assert */
      rc = -3;
      goto done;
    }
    if (!is_last) st.dict = NULL; /* LD:1917; window re-base / pruning: Q6, no effect */
  }
done:
  free(tok);
  free(hdr);
  free(st.stream_prev);
  free(st.dict_prev);
  if (st.stream_map.keys) trimap_free(&st.stream_map);
  if (st.dict_map.keys) trimap_free(&st.dict_map);
  return rc;
}

size_t ldo_compress_bound(size_t n) {
  size_t nblocks = 1;
  if (n > 65535) nblocks += (n - 65535 + 32767) / 32768;
  /* forced fixed/dynamic strategies can exceed the stored size: <= ~9.2 bits per
   * literal plus a dynamic header per block */
  return n + n / 4 + 512 * nblocks + 64;
}

/* LD:1960-1966, LD:1970-2024 */
int ldo_compress(const uint8_t *in, size_t n, int zlib, int level, int strategy, const uint8_t *dict,
                 size_t dict_len, uint8_t *out, size_t cap, size_t *out_len, int *pad_bits) {
  Writer w;
  int rc, pad;
  memset(&w, 0, sizeof(w));
  w.out = out;
  w.cap = cap;
  if (level > 9) return -4;
  if (!dict) dict_len = 0;
  if (zlib) {
    unsigned cmf = 7 * 16 + 8;
    unsigned fdict = dict ? 1 : 0;
    unsigned flg = 2 * 64 + fdict * 32;
    flg += 31 - (cmf * 256 + flg) % 31;
    write_bits(&w, cmf, 8);
    write_bits(&w, flg, 8);
    if (fdict) {
      uint32_t ad = ldo_adler32(dict, dict_len);
      write_bits(&w, (ad >> 24) & 0xFF, 8);
      write_bits(&w, (ad >> 16) & 0xFF, 8);
      write_bits(&w, (ad >> 8) & 0xFF, 8);
      write_bits(&w, ad & 0xFF, 8);
    }
  }
  rc = deflate_stream(&w, in, (long)n, level, strategy, dict, (long)dict_len, NULL, NULL, 0, NULL);
  if (rc) return rc;
  if (zlib) {
    uint32_t ad;
    flush_pad(&w, 1); /* LD:2004 BYTE_BOUNDARY */
    ad = ldo_adler32(in, n);
    write_bits(&w, (ad >> 24) & 0xFF, 8);
    write_bits(&w, (ad >> 16) & 0xFF, 8);
    write_bits(&w, (ad >> 8) & 0xFF, 8);
    write_bits(&w, ad & 0xFF, 8);
  }
  flush_pad(&w, 0); /* LD:1963 / 2021 OUTPUT */
  pad = (int)((8 - w.total_bitlen % 8) % 8);
  if (out_len) *out_len = w.len;
  if (pad_bits) *pad_bits = pad;
  return w.overflow ? -1 : 0;
}

long ldo_tokens(const uint8_t *in, size_t n, int level, const uint8_t *dict, size_t dict_len,
                uint16_t *tok_len, uint16_t *tok_dist, long cap) {
  Writer w;
  long cnt = 0;
  int rc;
  memset(&w, 0, sizeof(w));
  if (!dict) dict_len = 0;
  rc = deflate_stream(&w, in, (long)n, level, 0, dict, (long)dict_len, tok_len, tok_dist, cap, &cnt);
  return rc ? -1 : cnt;
}

/* ------------------------------------------------------------------ inflate */

/* LD:2127-2270 CreateReader, restated literally (4-byte refills, zero past end) */
typedef struct {
  const uint8_t *in;
  long n;
  long next; /* input_next_byte_pos, 1-based */
  long cache_bitlen;
  uint64_t cache;
} Reader;

static uint64_t r_load4(Reader *r) {
  uint64_t v = 0;
  int k;
  for (k = 0; k < 4; k++) {
    long p = r->next + k;
    if (p >= 1 && p <= r->n) v |= (uint64_t)r->in[p - 1] << (8 * k);
  }
  return v;
}
/* LD:2140-2161 */
static uint32_t read_bits(Reader *r, int bitlen) {
  uint64_t mask = (1ull << bitlen) - 1ull;
  uint32_t code;
  if (bitlen <= r->cache_bitlen) {
    code = (uint32_t)(r->cache & mask);
    r->cache >>= bitlen;
    r->cache_bitlen -= bitlen;
  } else {
    r->cache += r_load4(r) << r->cache_bitlen;
    r->next += 4;
    r->cache_bitlen += 32 - bitlen;
    code = (uint32_t)(r->cache & mask);
    r->cache >>= bitlen;
  }
  return code;
}
/* LD:2258-2260 */
static long bitlen_left(const Reader *r) { return (r->n - r->next + 1) * 8 + r->cache_bitlen; }
/* LD:2262-2267 */
static void skip_to_byte(Reader *r) {
  long s = r->cache_bitlen % 8;
  r->cache_bitlen -= s;
  r->cache >>= s;
}

typedef struct {
  int counts[17]; /* huffman_bitlen_counts[0..max] */
  int symbols[288];
  int min_bitlen;
} DecTable;

static unsigned rev_bits(unsigned v, int n) {
  unsigned r = 0;
  int k;
  for (k = 0; k < n; k++) {
    r = (r << 1) | (v & 1);
    v >>= 1;
  }
  return r;
}
/* LD:2206-2256 */
static int decode_sym(Reader *r, const DecTable *t) {
  long code = 0, first = 0, index = 0, count;
  int bitlen;
  const int min_bitlen = t->min_bitlen;
  if (min_bitlen > 0) {
    if (r->cache_bitlen < 15) {
      r->cache += r_load4(r) << r->cache_bitlen;
      r->next += 4;
      r->cache_bitlen += 32;
    }
    r->cache_bitlen -= min_bitlen;
    code = (long)(r->cache & ((1ull << min_bitlen) - 1ull));
    r->cache >>= min_bitlen;
    code = (long)rev_bits((unsigned)code, min_bitlen);
    count = t->counts[min_bitlen];
    if (code < count) return t->symbols[code];
    index = count;
    first = count * 2;
    code = code * 2;
  }
  for (bitlen = min_bitlen + 1; bitlen <= 15; bitlen++) {
    long bit = (long)(r->cache & 1ull), diff;
    r->cache >>= 1;
    r->cache_bitlen -= 1;
    code = (bit == 1) ? (code + 1 - code % 2) : code;
    count = t->counts[bitlen];
    diff = code - first;
    if (diff < count) return t->symbols[index + diff];
    index += count;
    first += count;
    first *= 2;
    code *= 2;
  }
  return -10;
}

/* LD:2303-2344 GetHuffmanForDecode: returns left (0 complete, >0 incomplete,
 * <0 over-subscribed). */
static int huffman_for_decode(const int *bitlens, int max_symbol, int max_bitlen, DecTable *t) {
  int offsets[17];
  int s, len, left;
  int min_bitlen = max_bitlen;
  for (len = 0; len <= 16; len++) t->counts[len] = 0;
  for (s = 0; s <= max_symbol; s++) {
    int b = bitlens[s];
    if (b > 0 && b < min_bitlen) min_bitlen = b;
    t->counts[b]++;
  }
  if (t->counts[0] == max_symbol + 1) { /* LD:2312-2314 */
    t->min_bitlen = 0;
    return 0;
  }
  left = 1;
  for (len = 1; len <= max_bitlen; len++) {
    left *= 2;
    left -= t->counts[len];
    if (left < 0) return left;
  }
  offsets[1] = 0;
  for (len = 1; len < max_bitlen; len++) offsets[len + 1] = offsets[len] + t->counts[len];
  for (s = 0; s <= max_symbol; s++) {
    int b = bitlens[s];
    if (b != 0) t->symbols[offsets[b]++] = s;
  }
  t->min_bitlen = min_bitlen;
  return left;
}

typedef struct {
  Reader r;
  uint8_t *out;
  size_t cap;
  uint64_t flushed;   /* bytes moved to result_buffer (LD:2432-2439) */
  long buffer_size;   /* state.buffer_size */
  int dict_loaded;    /* buffer[0] ~= nil (LD:2367) */
  const uint8_t *dict;
  long dlen;
  int too_small;
} Inflater;

static void put_byte(Inflater *s, unsigned b) {
  uint64_t pos = s->flushed + (uint64_t)s->buffer_size; /* 0-based slot for buffer[buffer_size+1] */
  if (pos < s->cap)
    s->out[pos] = (uint8_t)b;
  else
    s->too_small = 1;
  s->buffer_size++;
}
static unsigned get_buffer(const Inflater *s, long idx) { /* buffer[idx], idx may be <= 0 */
  if (idx >= 1) {
    uint64_t pos = s->flushed + (uint64_t)idx - 1;
    return pos < s->cap ? s->out[pos] : 0;
  }
  return s->dict[s->dlen + idx - 1]; /* LD:2374-2376 */
}
static void maybe_flush(Inflater *s) { /* LD:2432-2439, 2483-2488 */
  if (s->buffer_size >= 65536) {
    s->flushed += 32768;
    s->buffer_size -= 32768;
  }
}

/* LD:2353-2445 */
static int decode_until_eob(Inflater *s, const DecTable *lt, const DecTable *dt) {
  long buffer_end = 1;
  int symbol;
  if (s->dict && !s->dict_loaded) { /* LD:2367-2377 */
    buffer_end = -s->dlen + 1;
    s->dict_loaded = 1;
  }
  do {
    symbol = decode_sym(&s->r, lt);
    if (symbol < 0 || symbol > 285) return -10;
    if (symbol < 256) {
      put_byte(s, (unsigned)symbol);
    } else if (symbol > 256) {
      int sym = symbol - 256, dsym, k;
      long len = LEN_BASE[sym - 1], dist, cbi;
      if (sym >= 8) len += read_bits(&s->r, LEN_EXTRA[sym - 1]);
      dsym = decode_sym(&s->r, dt);
      if (dsym < 0 || dsym > 29) return -10;
      dist = DIST_BASE[dsym];
      if (dist > 4) dist += read_bits(&s->r, DIST_EXTRA[dsym]);
      cbi = s->buffer_size - dist + 1;
      if (cbi < buffer_end) return -11;
      if (cbi >= -257) {
        for (k = 0; k < len; k++) {
          put_byte(s, get_buffer(s, cbi));
          cbi++;
        }
      } else {
        cbi = s->dlen + cbi; /* 1-based index into dict_string_table */
        for (k = 0; k < len; k++) {
          put_byte(s, s->dict[cbi - 1]);
          cbi++;
        }
      }
    }
    if (bitlen_left(&s->r) < 0) return 2;
    maybe_flush(s);
  } while (symbol != 256);
  return 0;
}

/* LD:2450-2491 */
static int inflate_store(Inflater *s) {
  Reader *r = &s->r;
  long bytelen, comp, from_cache, k;
  skip_to_byte(r);
  bytelen = read_bits(r, 16);
  if (bitlen_left(r) < 0) return 2;
  comp = read_bits(r, 16);
  if (bitlen_left(r) < 0) return 2;
  if (bytelen % 256 + comp % 256 != 255) return -2;
  if (bytelen / 256 + comp / 256 != 255) return -2;
  /* LD:2170-2193 ReadBytes */
  from_cache = (r->cache_bitlen / 8 < bytelen) ? (r->cache_bitlen / 8) : bytelen;
  for (k = 0; k < from_cache; k++) {
    put_byte(s, (unsigned)(r->cache & 0xFF));
    r->cache >>= 8;
  }
  r->cache_bitlen -= from_cache * 8;
  bytelen -= from_cache;
  if ((r->n - r->next - bytelen + 1) * 8 + r->cache_bitlen < 0) return 2;
  for (k = 0; k < bytelen; k++) put_byte(s, r->in[r->next - 1 + k]);
  r->next += bytelen;
  maybe_flush(s);
  return 0;
}

static DecTable FIX_LT, FIX_DT;
static int fix_dec_ready = 0;
static void fix_dec_init(void) { /* LD:2826-2841 */
  int bl[288], i;
  if (fix_dec_ready) return;
  for (i = 0; i < 288; i++) bl[i] = fix_l_bitlen(i);
  huffman_for_decode(bl, 287, 9, &FIX_LT);
  FIX_LT.min_bitlen = 7; /* LD:2498 */
  for (i = 0; i < 32; i++) bl[i] = 5;
  huffman_for_decode(bl, 31, 5, &FIX_DT);
  FIX_DT.min_bitlen = 5;
  fix_dec_ready = 1;
}

/* LD:2506-2615 */
static int inflate_dynamic(Inflater *s) {
  Reader *r = &s->r;
  int nlen = (int)read_bits(r, 5) + 257;
  int ndist = (int)read_bits(r, 5) + 1;
  int ncode = (int)read_bits(r, 4) + 4;
  int cl[19], lbl[288], dbl[32], i, index, err;
  DecTable ct, lt, dt;
  if (nlen > 286 || ndist > 30) return -3;
  for (i = 0; i < 19; i++) cl[i] = 0;
  for (i = 1; i <= ncode; i++) cl[CL_ORDER[i - 1]] = (int)read_bits(r, 3);
  err = huffman_for_decode(cl, 18, 7, &ct);
  if (err != 0) return -4;
  for (i = 0; i < 288; i++) lbl[i] = 0;
  for (i = 0; i < 32; i++) dbl[i] = 0;
  index = 0;
  while (index < nlen + ndist) {
    int symbol = decode_sym(r, &ct), bitlen;
    if (symbol < 0) return symbol;
    if (symbol < 16) {
      if (index < nlen)
        lbl[index] = symbol;
      else
        dbl[index - nlen] = symbol;
      index++;
    } else {
      bitlen = 0;
      if (symbol == 16) {
        if (index == 0) return -5;
        bitlen = (index - 1 < nlen) ? lbl[index - 1] : dbl[index - nlen - 1];
        symbol = 3 + (int)read_bits(r, 2);
      } else if (symbol == 17) {
        symbol = 3 + (int)read_bits(r, 3);
      } else {
        symbol = 11 + (int)read_bits(r, 7);
      }
      if (index + symbol > nlen + ndist) return -6;
      while (symbol > 0) {
        symbol--;
        if (index < nlen)
          lbl[index] = bitlen;
        else
          dbl[index - nlen] = bitlen;
        index++;
      }
    }
  }
  if (lbl[256] == 0) return -9;
  err = huffman_for_decode(lbl, nlen - 1, 15, &lt);
  if (err != 0 && (err < 0 || nlen != lt.counts[0] + lt.counts[1])) return -7;
  err = huffman_for_decode(dbl, ndist - 1, 15, &dt);
  if (err != 0 && (err < 0 || ndist != dt.counts[0] + dt.counts[1])) return -8;
  return decode_until_eob(s, &lt, &dt);
}

/* LD:2620-2644 */
static int inflate_blocks(Inflater *s) {
  int is_last = 0;
  fix_dec_init();
  while (!is_last) {
    int type, status;
    is_last = (read_bits(&s->r, 1) == 1);
    type = (int)read_bits(&s->r, 2);
    if (type == 0)
      status = inflate_store(s);
    else if (type == 1)
      status = decode_until_eob(s, &FIX_LT, &FIX_DT);
    else if (type == 2)
      status = inflate_dynamic(s);
    else
      return -1;
    if (status != 0) return status;
  }
  return 0;
}

/* LD:2648-2656, LD:2660-2727 */
int ldo_decompress(const uint8_t *in, size_t n, int zlib, const uint8_t *dict, size_t dict_len,
                   uint8_t *out, size_t cap, size_t *out_len, size_t *bytes_left) {
  Inflater s;
  int status;
  long left;
  memset(&s, 0, sizeof(s));
  s.r.in = in;
  s.r.n = (long)n;
  s.r.next = 1;
  s.out = out;
  s.cap = out ? cap : 0;
  s.dict = dict;
  s.dlen = dict ? (long)dict_len : 0;
  if (out_len) *out_len = 0;
  if (bytes_left) *bytes_left = 0;
  if (zlib) {
    unsigned cmf = read_bits(&s.r, 8), flg;
    if (bitlen_left(&s.r) < 0) return 2;
    if (cmf % 16 != 8) return -12;
    if (cmf / 16 > 7) return -13;
    flg = read_bits(&s.r, 8);
    if (bitlen_left(&s.r) < 0) return 2;
    if ((cmf * 256 + flg) % 31 != 0) return -14;
    if ((flg / 32) % 2 == 1) {
      uint32_t b3, b2, b1, b0, actual;
      if (!dict) return -16;
      b3 = read_bits(&s.r, 8);
      b2 = read_bits(&s.r, 8);
      b1 = read_bits(&s.r, 8);
      b0 = read_bits(&s.r, 8);
      actual = b3 * 16777216u + b2 * 65536u + b1 * 256u + b0;
      if (bitlen_left(&s.r) < 0) return 2;
      if (actual != ldo_adler32(dict, dict_len)) return -17;
    }
  }
  status = inflate_blocks(&s);
  if (status != 0) return status;
  if (out_len) *out_len = (size_t)(s.flushed + (uint64_t)s.buffer_size);
  if (zlib) {
    uint32_t a0, a1, a2, a3, expected;
    skip_to_byte(&s.r);
    a0 = read_bits(&s.r, 8);
    a1 = read_bits(&s.r, 8);
    a2 = read_bits(&s.r, 8);
    a3 = read_bits(&s.r, 8);
    if (bitlen_left(&s.r) < 0) return 2;
    expected = a0 * 16777216u + a1 * 65536u + a2 * 256u + a3;
    if (s.too_small) return out ? 1 : 0;
    if (expected != ldo_adler32(out, (size_t)(s.flushed + (uint64_t)s.buffer_size))) return -15;
  }
  if (s.too_small) return out ? 1 : 0;
  left = bitlen_left(&s.r);
  if (bytes_left) *bytes_left = (size_t)((left - left % 8) / 8);
  return 0;
}

/* ------------------------------------------------------------------ print codec */

/* LD:3165-3230 */
static const char P6[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789()";

/* LD:3311-3353 */
size_t ldo_print6_encode(const uint8_t *in, size_t n, uint8_t *out) {
  size_t i = 0, o = 0;
  uint32_t cache = 0;
  int cache_bitlen = 0;
  while (i + 2 < n) {
    uint32_t c = (uint32_t)in[i] + (uint32_t)in[i + 1] * 256u + (uint32_t)in[i + 2] * 65536u;
    i += 3;
    out[o++] = (uint8_t)P6[c & 63];
    out[o++] = (uint8_t)P6[(c >> 6) & 63];
    out[o++] = (uint8_t)P6[(c >> 12) & 63];
    out[o++] = (uint8_t)P6[(c >> 18) & 63];
  }
  while (i < n) {
    cache += (uint32_t)in[i] << cache_bitlen;
    cache_bitlen += 8;
    i++;
  }
  while (cache_bitlen > 0) {
    out[o++] = (uint8_t)P6[cache & 63];
    cache >>= 6;
    cache_bitlen -= 6;
  }
  return o;
}

static int p6_value(unsigned c) { /* LD:3232-3297 */
  if (c >= 'a' && c <= 'z') return (int)(c - 'a');
  if (c >= 'A' && c <= 'Z') return (int)(c - 'A') + 26;
  if (c >= '0' && c <= '9') return (int)(c - '0') + 52;
  if (c == '(') return 62;
  if (c == ')') return 63;
  return -1;
}
static int is_cntrl_or_space(unsigned c) { return c < 32 || c == 127 || c == ' '; } /* "[%c ]" */

/* LD:3365-3417 */
int ldo_print6_decode(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len) {
  size_t b = 0, e = n, i, o = 0;
  uint32_t cache = 0;
  int cache_bitlen = 0;
  while (b < e && is_cntrl_or_space(in[b])) b++;
  while (e > b && is_cntrl_or_space(in[e - 1])) e--;
  if (e - b == 1) return -1;
  i = b;
  while (i + 3 < e) {
    int x1 = p6_value(in[i]), x2 = p6_value(in[i + 1]), x3 = p6_value(in[i + 2]),
        x4 = p6_value(in[i + 3]);
    uint32_t c;
    if (x1 < 0 || x2 < 0 || x3 < 0 || x4 < 0) return -1;
    i += 4;
    c = (uint32_t)x1 + (uint32_t)x2 * 64u + (uint32_t)x3 * 4096u + (uint32_t)x4 * 262144u;
    out[o++] = (uint8_t)(c & 255);
    out[o++] = (uint8_t)((c >> 8) & 255);
    out[o++] = (uint8_t)((c >> 16) & 255);
  }
  while (i < e) {
    int x = p6_value(in[i]);
    if (x < 0) return -1;
    cache += (uint32_t)x << cache_bitlen;
    cache_bitlen += 6;
    i++;
  }
  while (cache_bitlen >= 8) {
    out[o++] = (uint8_t)(cache & 255);
    cache >>= 8;
    cache_bitlen -= 8;
  }
  *out_len = o;
  return 0;
}


/* ======================================================================= CreateCodec */

/* LD:2921-3056.  The pattern strings of the reference are replaced by what they denote: sets of
 * bytes and translate tables; gsub's left-to-right, non-overlapping scan is kept literally. */
int ldo_codec_build(const uint8_t *reserved, size_t nr, const uint8_t *escape, size_t ne,
                    const uint8_t *map, size_t nm, ldo_codec *c, const char **err) {
  static const char *e_noesc = "No escape characters supplied.";
  static const char *e_count = "The number of reserved characters must be at least as many as the number of mapped chars.";
  static const char *e_none = "No characters to encode.";
  static const char *e_dup = "There must be no duplicate characters in the concatenation of reserved_chars, escape_chars and map_chars.";
  static const char *e_out = "Out of escape characters.";
  *err = NULL;
  if (ne == 0) { *err = e_noesc; return -1; }              /* LD:2928 */
  if (nr < nm) { *err = e_count; return -1; }              /* LD:2929-2932 */
  if (nr == 0) { *err = e_none; return -1; }               /* LD:2933 */
  size_t nall = nr + ne + nm;
  uint8_t *all = (uint8_t *)malloc(nall ? nall : 1);
  if (!all) { *err = "out of memory"; return -1; }
  memcpy(all, reserved, nr);
  memcpy(all + nr, escape, ne);
  memcpy(all + nr + ne, map, nm);
  int taken[256] = {0}, translated[256] = {0};
  for (size_t i = 0; i < nall; i++) {                      /* LD:2937-2946 */
    if (taken[all[i]]) { free(all); *err = e_dup; return -1; }
    taken[all[i]] = 1;
  }
  memset(c, 0, sizeof(*c));
  for (int b = 0; b < 256; b++) {
    c->enc_len[b] = 1;
    c->enc0[b] = (uint8_t)b;
    c->map_back[b] = (uint8_t)b;
    for (int p = 0; p < 256; p++) c->pass_dec[p][b] = 0xFFFF;
  }
  for (size_t i = 0; i < nr; i++) c->reserved[reserved[i]] = 1;   /* LD:3040 */
  if (nm > 0) {                                            /* LD:2957-2971 */
    c->has_map = 1;
    for (size_t i = 0; i < nm; i++) {
      c->enc0[reserved[i]] = map[i];
      translated[reserved[i]] = 1;
      c->map_back[map[i]] = reserved[i];
    }
  }
  size_t esc_index = 0;                                    /* LD:2973-2977 */
  int r = 0;
  for (size_t i = 0; i < nall; i++) {                      /* LD:2981-3022 */
    const uint8_t ch = all[i];
    if (!translated[ch]) {
      while (r >= 256 || taken[r]) {
        r++;
        if (r > 255) {                                     /* switch to the next escape char */
          c->pass_escape[c->npass++] = escape[esc_index];
          esc_index++;
          r = 0;
          if (esc_index >= ne) { free(all); *err = e_out; return -1; }   /* LD:2999-3005 */
        }
      }
      c->enc_len[ch] = 2;
      c->enc0[ch] = escape[esc_index];
      c->enc1[ch] = (uint8_t)r;
      translated[ch] = 1;
      c->pass_dec[c->npass][r] = ch;
      r++;
    }
    if (i + 1 == nall) c->pass_escape[c->npass++] = escape[esc_index];   /* LD:3016-3021 */
  }
  free(all);
  return 0;
}

size_t ldo_codec_encode(const ldo_codec *c, const uint8_t *in, size_t n, uint8_t *out) {  /* LD:3030-3037 */
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    out[o++] = c->enc0[in[i]];
    if (c->enc_len[in[i]] == 2) out[o++] = c->enc1[in[i]];
  }
  return o;
}

int ldo_codec_decode(const ldo_codec *c, const uint8_t *in, size_t n, uint8_t *out, size_t *out_len) {  /* LD:3042-3053 */
  for (size_t i = 0; i < n; i++)
    if (c->reserved[in[i]]) return -1;                     /* LD:3048 */
  uint8_t *a = (uint8_t *)malloc(n ? n : 1), *b = (uint8_t *)malloc(n ? n : 1);
  if (!a || !b) { free(a); free(b); return -1; }
  size_t len = n;
  for (size_t i = 0; i < n; i++) a[i] = c->has_map ? c->map_back[in[i]] : in[i];   /* first gsub */
  for (int p = 0; p < c->npass; p++) {                     /* one gsub per escape character, in order */
    const uint8_t E = c->pass_escape[p];
    size_t o = 0, i = 0;
    while (i < len) {
      if (a[i] == E && i + 1 < len && c->pass_dec[p][a[i + 1]] != 0xFFFF) {
        b[o++] = (uint8_t)c->pass_dec[p][a[i + 1]];
        i += 2;
      } else {
        b[o++] = a[i++];
      }
    }
    uint8_t *t = a; a = b; b = t;
    len = o;
  }
  memcpy(out, a, len);
  *out_len = len;
  free(a);
  free(b);
  return 0;
}
