// k_inflate.h -- north-star part (5): batched Inflate, one stream per wavefront, decode
// tables resident in LDS.
//
// Replaces CreateReader (LD:2127-2270), GetHuffmanForDecode (LD:2303-2344),
// DecodeUntilEndOfBlock (LD:2353-2445), DecompressStoreBlock/FixBlock/DynamicBlock
// (LD:2450-2615), Inflate (LD:2620-2644) and the framing checks of
// DecompressDeflateInternal / DecompressZlibInternal (LD:2648-2727).
//
// Reader model: the reference refills 4 bytes at a time and returns zero bits past the
// end of input; ReaderBitlenLeft() = 8*len - bits consumed, whatever the refill pattern.
// The kernel keeps `consumed` and reproduces the same late exhaustion checks (Q11).
// The bit-serial state is wave-uniform; lanes cooperate on table construction and on
// match copies.
#pragma once
#include "ldh_common.h"

namespace ldh {

constexpr int kLutBits = 10;
constexpr int kLutSize = 1 << kLutBits;

// LDS per wavefront decides how many of a batch's streams are resident at once (one stream = one
// wavefront, 32 wavefronts per CU at most), so the tables are sized per use: literal/length codes get
// a 10-bit lookup and 288 symbols, distance (and code-length) codes a 9-bit lookup and 32 symbols.
template <int LUT_BITS, int NSYM>
struct DecTabT {
  static constexpr int kBits = LUT_BITS;
  static constexpr int kSize = 1 << LUT_BITS;
  uint16_t lut[kSize];     // entry for codes of <= kBits bits (see lut_entry)
  uint16_t symbols[NSYM];  // puff order: by (length, symbol)
  uint32_t counts[16];
  uint32_t offs[16];
  uint32_t first[16];      // first canonical code of each length
  uint32_t running[16];
};
using LitTab = DecTabT<kLutBits, 288>;
using DistTab = DecTabT<9, 32>;

// Lookup entries carry what the lane-parallel decoder needs, ready to use (computing base and
// extra-bit count of LD:188-240 from the symbol cost a fifth of its vector instructions):
//   kLutPlain   (sym << 4) | bits; 0 = not in the table                                  code-length alphabet
//   kLutLitLen  literal:        byte << 4 | bits
//               length symbol:  1 << 15 | (base - 3) << 7 | extra_bits << 4 | bits
//               end of block:   kLutEob | bits;   286, 287:  kLutBad | bits
//   kLutDist    mant << 8 | extra_bits << 4 | bits:  dist = 1 + (mant << eb) + extra;   30, 31:  kLutBad | bits
//   both packed kinds: kLutMiss = the code is longer than the table's index (or no code at all): the canonical walk.
// Bits 14:13 of an entry that is not a length are its class -- 0 a token the walk passes, else one it cannot -- so that
// "can the walk pass" is one AND and one compare on (is length ? distance entry : literal entry).
enum { kLutPlain = 0, kLutLitLen = 1, kLutDist = 2 };
constexpr uint32_t kLutEob = 0x2000u, kLutBad = 0x4000u, kLutMiss = 0x6000u, kLutClass = 0x6000u;
__device__ inline uint32_t lut_entry(int mode, uint32_t sym, uint32_t bits) {
  if (mode == kLutLitLen) {
    if (sym > 285) return kLutBad | bits;
    if (sym == 256) return kLutEob | bits;
    if (sym > 256) {
      const uint32_t ls = sym - 257;
      const uint32_t eb = (ls < 8 || ls == 28) ? 0u : (ls >> 2) - 1;
      const uint32_t b3 = ls < 8 ? ls : (ls == 28 ? 255u : (4u + (ls & 3)) << eb);
      return (1u << 15) | (b3 << 7) | (eb << 4) | bits;
    }
  } else if (mode == kLutDist) {
    if (sym > 29) return kLutBad | bits;
    const uint32_t eb = sym < 4 ? 0u : (sym >> 1) - 1;
    const uint32_t mant = sym < 4 ? sym : 2u + (sym & 1);
    return (mant << 8) | (eb << 4) | bits;
  }
  return (sym << 4) | bits;
}

struct InflateArgs {
  const uint8_t *in;
  const uint64_t *in_off;   // stream s = in[in_off[s] .. in_off[s+1])            (in_len == nullptr)
  const uint32_t *in_len;   // or: stream s = in + s*in_stride, in_len[s] bytes  (slots)
  uint64_t in_stride;
  uint32_t count;
  int fmt;
  const uint8_t *dict;
  uint32_t dlen;
  uint32_t dict_adler;
  uint8_t *out;
  uint64_t out_stride;
  uint32_t *out_len;
  uint32_t *bytes_left;
  int32_t *status;
  uint32_t *expect_adler;  // zlib trailer value per stream (checked by k_zlib_check)
};

struct BitReader {
  const uint32_t *base;  // 4-byte aligned pointer at or before the stream's first byte
  uint32_t skip;         // bytes between base and the first byte
  uint32_t nbytes;       // stream length
  uint32_t chunk;        // per lane: word chunk_w0 + lane
  uint32_t chunk_w0;     // uniform: first word index held in `chunk`
  uint32_t next_w;       // uniform: next word to append to bitbuf
  uint64_t bitbuf;       // uniform
  uint32_t bitcnt;       // uniform: valid bits in bitbuf
  uint64_t consumed;     // uniform: bits consumed from the stream so far
};

__device__ inline void br_load_chunk(BitReader &R, uint32_t w0, uint32_t lane) {
  const uint32_t w = w0 + lane;
  const uint64_t first = (uint64_t)w * 4;            // byte offset from base
  const uint64_t end = (uint64_t)R.skip + R.nbytes;  // one past the last valid byte
  uint32_t x = 0;
  if (first < end) {
    x = R.base[w];
    if (first + 4 > end) x &= (1u << (8 * (uint32_t)(end - first))) - 1u;  // zeros past the end
  }
  R.chunk = x;
  R.chunk_w0 = w0;
}
__device__ inline void br_refill(BitReader &R, uint32_t lane) {
  while (R.bitcnt <= 32) {
    if (R.next_w - R.chunk_w0 >= 64) br_load_chunk(R, R.next_w, lane);
    const uint32_t idx = __builtin_amdgcn_readfirstlane(R.next_w - R.chunk_w0);
    const uint32_t word = __builtin_amdgcn_readlane(R.chunk, idx);
    R.bitbuf |= (uint64_t)word << R.bitcnt;
    R.bitcnt += 32;
    R.next_w++;
  }
}
// position the reader at stream byte `byte_pos`
__device__ inline void br_seek(BitReader &R, uint64_t byte_pos, uint32_t lane) {
  const uint64_t abs = (uint64_t)R.skip + byte_pos;
  R.consumed = byte_pos * 8;
  R.next_w = (uint32_t)(abs >> 2);
  R.bitbuf = 0;
  R.bitcnt = 0;
  br_load_chunk(R, R.next_w, lane);
  br_refill(R, lane);
  const uint32_t drop = 8 * (uint32_t)(abs & 3);
  R.bitbuf >>= drop;
  R.bitcnt -= drop;
}
__device__ inline void br_init(BitReader &R, const uint8_t *p, uint32_t n, uint32_t lane) {
  const uintptr_t a = (uintptr_t)p;
  R.base = (const uint32_t *)(a & ~(uintptr_t)3);
  R.skip = (uint32_t)(a & 3);
  R.nbytes = n;
  br_seek(R, 0, lane);
}
// ReadBits (LD:2140-2161); n <= 16
__device__ inline uint32_t br_read(BitReader &R, uint32_t n, uint32_t lane) {
  br_refill(R, lane);
  const uint32_t v = (uint32_t)(R.bitbuf & ((1ull << n) - 1ull));
  R.bitbuf >>= n;
  R.bitcnt -= n;
  R.consumed += n;
  return v;
}
__device__ inline int64_t br_left(const BitReader &R) {  // ReaderBitlenLeft (LD:2258-2260)
  return (int64_t)R.nbytes * 8 - (int64_t)R.consumed;
}
__device__ inline void br_align(BitReader &R, uint32_t lane) {  // SkipToByteBoundary (LD:2262-2267)
  const uint32_t s = (uint32_t)((8 - (R.consumed & 7)) & 7);
  br_refill(R, lane);
  R.bitbuf >>= s;
  R.bitcnt -= s;
  R.consumed += s;
}

// GetHuffmanForDecode (LD:2303-2344) over lens[0..nsym) (LDS).  Returns `left`:
// 0 complete (also when there is no code at all, LD:2312-2314), > 0 incomplete,
// < 0 over-subscribed.  Called by the whole wavefront (block = one wavefront).
// W: wavefronts of the workgroup that decodes this stream (1, or more for k_inflate_wide).  With W > 1 every wavefront
// runs the same wave-uniform code on the same data (so all of them take the same branches and meet at the same
// barriers) and wavefront 0 alone makes the stores to LDS that must happen once.
template <class DecTab, int W = 1>
__device__ int build_table(DecTab &T, const uint8_t *lens, int nsym, int max_bitlen, int lut_bits,
                           uint32_t tid, int mode) {
  const uint32_t lane = tid & 63;
  const bool writer = W == 1 || tid < 64;
  if (tid < 16) {
    T.counts[tid] = 0;
    T.running[tid] = 0;
    T.offs[tid] = 0;
    T.first[tid] = 0;
  }
  for (int i = tid; i < DecTab::kSize; i += W * kWave) T.lut[i] = mode == kLutPlain ? 0 : (uint16_t)kLutMiss;
  __syncthreads();
  for (int base = 0; base < nsym; base += kWave) {
    const int i = base + (int)lane;
    if (writer && i < nsym) atomicAdd(&T.counts[lens[i]], 1u);
  }
  __syncthreads();
  if (T.counts[0] == (uint32_t)nsym) return 0;  // no codes: "complete", every decode fails
  int left = 1;
  for (int len = 1; len <= max_bitlen; len++) {
    left = left * 2 - (int)T.counts[len];
    if (left < 0) return left;
  }
  if (tid == 0) {
    uint32_t off = 0, code = 0;
    for (int len = 1; len <= 15; len++) {
      code = (code + (len > 1 ? T.counts[len - 1] : 0u)) << 1;
      T.first[len] = code;
      T.offs[len] = off;
      off += T.counts[len];
    }
  }
  __syncthreads();
  for (int base = 0; base < nsym; base += kWave) {
    const int i = base + (int)lane;
    const uint32_t l = i < nsym ? lens[i] : 0;
    const bool active = l != 0;
    unsigned long long mask = ballot64(active);
#pragma unroll
    for (int bit = 0; bit < 4; bit++) {
      const unsigned long long bal = ballot64(active && ((l >> bit) & 1));
      mask &= ((l >> bit) & 1) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (active && writer) {
      const uint32_t in_len = T.running[l] + rank;
      T.symbols[T.offs[l] + in_len] = (uint16_t)i;
      if ((int)l <= lut_bits) {
        const uint32_t c = T.first[l] + in_len;
        const uint32_t r = __brev(c) >> (32 - l);
        const uint16_t ent = (uint16_t)lut_entry(mode, (uint32_t)i, l);
        for (uint32_t k = r; k < (1u << lut_bits); k += (1u << l)) T.lut[k] = ent;
      }
    }
    __syncthreads();
    if (active && writer && rank == 0) T.running[l] += (uint32_t)__popcll(mask);
    __syncthreads();
  }
  return left;
}

// Decode (LD:2206-2256).  Returns the symbol or -10; consumes its bits.
template <class DecTab>
__device__ inline int decode_sym(const DecTab &T, int lut_bits, BitReader &R, uint32_t lane) {
  br_refill(R, lane);
  // lut_bits 0: a table with packed entries (kLutLitLen / kLutDist); take the canonical walk
  const uint32_t e =
      lut_bits ? __builtin_amdgcn_readfirstlane((uint32_t)T.lut[R.bitbuf & ((1u << lut_bits) - 1u)]) : 0u;
  int sym;
  uint32_t len;
  if (e) {
    sym = (int)(e >> 4);
    len = e & 15;
  } else {  // canonical bit-serial walk (puff), also the path of every invalid code
    uint64_t bits = R.bitbuf;
    int code = 0, first = 0, index = 0;
    sym = -10;
    len = 15;
    for (int l = 1; l <= 15; l++) {
      code |= (int)(bits & 1);
      bits >>= 1;
      const int count = (int)__builtin_amdgcn_readfirstlane(T.counts[l]);
      if (code - first < count) {
        sym = (int)__builtin_amdgcn_readfirstlane((uint32_t)T.symbols[index + (code - first)]);
        len = (uint32_t)l;
        break;
      }
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
  }
  R.bitbuf >>= len;
  R.bitcnt -= len;
  R.consumed += len;
  return sym;
}

struct OutState {
  uint8_t *out;
  uint64_t cap;
  uint64_t opos;     // bytes produced
  const uint8_t *dict;
  uint32_t dlen;
  bool dict_loaded;  // buffer[0] ~= nil (LD:2367)
};

// One token of DecodeUntilEndOfBlock (LD:2379-2440), wave-uniform ("scalar") form.  Used for
// tokens the lane-parallel decoder cannot resolve (codes longer than the lookup table, invalid
// codes).  *eob is set when the token was the end-of-block symbol.
// The reference's buffer_size only differs from the output position after it reached 65536,
// where no distance (<= 32768) can fail, so `dist > opos + ext` is the -11 test (LD:2406-2410).
template <int W = 1>
__device__ int decode_one(const LitTab &LT, const DistTab &DT, int lbits, int dbits, BitReader &R,
                          OutState &O, uint32_t ext, uint32_t tid, bool *eob) {
  const uint32_t lane = tid & 63;
  const int symbol = decode_sym(LT, lbits, R, lane);
  *eob = symbol == 256;
  if (symbol < 0 || symbol > 285) return -10;
  if (symbol < 256) {
    if (O.opos >= O.cap) return kStatusOutputFull;
    if (tid == 0) O.out[O.opos] = (uint8_t)symbol;
    O.opos++;
  } else if (symbol > 256) {
    const uint32_t ls = (uint32_t)symbol - 257;
    uint32_t len;  // LD:188-200 base lengths, computed instead of tabulated
    if (ls < 8) {
      len = 3 + ls;
    } else if (ls == 28) {
      len = 258;
    } else {
      const uint32_t eb = (ls >> 2) - 1;
      len = 3 + ((4u + (ls & 3)) << eb) + br_read(R, eb, lane);
    }
    const int dsym = decode_sym(DT, dbits, R, lane);
    if (dsym < 0 || dsym > 29) return -10;
    uint32_t dist;
    if (dsym < 4) {
      dist = (uint32_t)dsym + 1;
    } else {
      const uint32_t eb = dist_extra_bits((uint32_t)dsym);
      dist = 1 + ((2u + ((uint32_t)dsym & 1)) << eb) + br_read(R, eb, lane);
    }
    if ((uint64_t)dist > O.opos + ext) return -11;
    if (O.opos + len > O.cap) return kStatusOutputFull;
    // bytes this workgroup stored earlier must be visible to the loads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (W > 1) __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int64_t src0 = (int64_t)O.opos - (int64_t)dist;
    for (uint32_t k = tid; k < len; k += W * kWave) {
      const uint32_t j = dist >= len ? k : (k % dist);
      const int64_t si = src0 + j;
      const uint8_t byte = si < 0 ? O.dict[(int64_t)O.dlen + si] : O.out[si];
      O.out[O.opos + k] = byte;
    }
    O.opos += len;
  }
  if (br_left(R) < 0) return 2;  // LD:2428
  return 0;
}

// a value the caller knows to be the same in every lane, moved to scalar registers
__device__ inline uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); }
__device__ inline uint64_t uni64(uint64_t x) {
  return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x);
}

__device__ inline void br_seek_bits(BitReader &R, uint64_t bit_pos, uint32_t lane) {
  br_seek(R, bit_pos >> 3, lane);
  const uint32_t drop = (uint32_t)(bit_pos & 7);
  R.bitbuf >>= drop;
  R.bitcnt -= drop;
  R.consumed = bit_pos;
}

// What the token that would start at one bit offset decodes to, speculatively (DecodeUntilEndOfBlock's table look-ups,
// LD:2379-2440, for one lane): w0..w2 are the three dwords of input at the offset's dword, sh its bit in the first.
// Values of the other kind of token are computed regardless and are garbage (every index stays inside its table).
struct SpecTok {
  uint32_t e, e2;      // the literal/length entry and the distance entry
  uint32_t he;         // the entry that says whether the walk can pass: is_len ? e2 : e
  bool is_len;         // a match
  bool hard;           // the walk cannot pass: end of block, a code the tables do not hold, an invalid symbol
  uint32_t mlen, dist; // the match (is_len && !hard)
  uint32_t tok_bits;   // bits of the token (of the literal/length code alone for one that is not a match)
  uint32_t out_len;    // bytes it produces (0 when hard)
  // the classes (used off the fast path only)
  __device__ uint32_t cls() const { return he & kLutClass; }
  __device__ bool is_lit() const { return !is_len & ((e & kLutClass) == 0); }
  __device__ bool is_eob() const { return !is_len & ((e & kLutClass) == kLutEob); }
  __device__ bool slow() const { return cls() == kLutMiss; }
  __device__ bool bad() const { return cls() == kLutBad; }
  __device__ uint32_t lit() const { return (e >> 4) & 255u; }
};
__device__ inline SpecTok spec_decode(const LitTab &LT, const DistTab &DT, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t sh) {
  SpecTok T;
  const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
  const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
  T.e = LT.lut[lo & (LitTab::kSize - 1)];
  const uint32_t l1 = T.e & 15;
  T.is_len = (T.e >> 15) != 0;
  const uint32_t leb = (T.e >> 4) & 7;  // (a table hit is at most 10 bits, 5 extra: all inside `lo`)
  T.mlen = 3 + ((T.e >> 7) & 255) + ((lo >> l1) & ((1u << leb) - 1u));
  const uint32_t t = l1 + leb;
  const uint32_t din = __builtin_amdgcn_alignbit(hi, lo, t);  // 32 bits from bit t on: 9 + 13 are needed at most
  T.e2 = DT.lut[din & (DistTab::kSize - 1)];
  const uint32_t l2 = T.e2 & 15, deb = (T.e2 >> 4) & 15;
  T.dist = 1 + (((T.e2 >> 8) & 3) << deb) + ((din >> l2) & ((1u << deb) - 1u));
  T.he = T.is_len ? T.e2 : T.e;
  T.hard = (T.he & kLutClass) != 0;
  T.tok_bits = T.is_len ? t + l2 + deb : l1;
  T.out_len = T.hard ? 0u : (T.is_len ? T.mlen : 1u);
  return T;
}

// DecodeUntilEndOfBlock (LD:2353-2445), lane-parallel.
// The scalar unit, not the vector lanes, bounded the wave-uniform decoder (every branch and
// every uniform bit operation is a scalar instruction, one per cycle per CU).  Here lane k
// decodes, speculatively, the token that would start k bits after the current position
// (table lookups are per-lane LDS gathers); walking next[k] = k + token_bits from lane 0
// then picks the real tokens of this 64-bit window (about eight per round).  Literals are
// stored by their lanes at once.  A match whose source is older than everything in flight
// is loaded at once (one lane per byte) and stored a round later, so that the load latency
// overlaps the next round's decode; matches that read recent output are copied in order.
// Tokens the tables cannot resolve fall back to decode_one.
// The scalar unit bounds this function (one instruction per cycle for the CU's four SIMDs), so
// wave-uniform state is kept narrow: output positions are 32-bit (k_inflate clamps the
// capacity below 2^32: lengths are reported as uint32 anyway), and the bit position is a
// 32-bit offset `rel` into the staged window plus a 64-bit base that changes only when the
// window is reloaded.
template <int COPY>  // (0: the copy that is called, 1: the copy that is inlined; see k_inflate below)
__device__ int decode_block(const LitTab &LT, const DistTab &DT, BitReader &R, OutState &O, uint32_t lane,
                            uint32_t *s_in /* LDS, 64 + 4 dwords */, uint32_t *s_tk /* LDS, 64 dwords */) {
  uint32_t ext = 0;
  if (O.dict && !O.dict_loaded) {  // LD:2367-2377
    ext = O.dlen;
    O.dict_loaded = true;
  }
  // (The called copy gets its state through references: the compiler cannot tell that what it loads from them is
  // wave-uniform and would keep positions, pointers and the loop's conditions in vector registers.  v_readfirstlane
  // says so; in the inlined copy, where they are scalar already, it folds away.)
  const uint32_t r_nbytes = uni(R.nbytes), r_skip = uni(R.skip);
  const uint32_t *const r_base = (const uint32_t *)uni64((uint64_t)(uintptr_t)R.base);
  const uint64_t nbits = (uint64_t)r_nbytes * 8;
  const uint64_t skip_bits = (uint64_t)r_skip * 8;
  const uint32_t cap = uni((uint32_t)O.cap);
  // (a global, not a generic, pointer: loads and stores take the scalar base and a 32-bit lane offset as they are,
  // without a 64-bit vector add per access, and do not count as LDS operations in the wait counters)
  typedef __attribute__((address_space(1))) uint8_t *gbytes_t;
  const gbytes_t out = (gbytes_t)uni64((uint64_t)(uintptr_t)O.out);
  uint32_t op = uni((uint32_t)O.opos);  // output position (uniform)
  uint64_t bp = uni64(R.consumed);      // bits consumed so far (uniform); current only while !have
  int64_t bp0 = 0;                 // bits consumed = bp0 + rel while the window is valid
  uint32_t rel = 0xFFFFFFFFu;      // bit offset of the current position in the window (all ones while there is none)
  uint32_t lim = 0;                // the input ends `lim` bits after the window's bit 0
  bool have = false;
  const unsigned long long le_mask = (2ull << lane) - 1ull;  // lanes 0..lane
  uint32_t pend_n = 0, pend_op = 0, pend_v = 0;  // the plain round whose bytes are not stored yet: count, position, bytes
  // (Fencing only in rounds whose match sources were not yet known to have landed was measured: 11.3 ms per GiB against
  // 11.0 with the fence every round; one 128 KiB string 5.05 against 4.81 ms.)
  for (;;) {
    // (The plain rounds are a loop of their own inside the loop over all rounds: what is carried from one round to the
    // next -- positions, the window's base and limit -- then keeps its registers; with one loop the paths through the
    // general case made the compiler shuffle a dozen scalar registers at every back edge.)
    SpecTok T;
    uint32_t bit, acc, k, my_off, my_pos;
    bool is_tok;
    for (;;) {
    if (rel >= 58 * 32) {  // (also: no window yet)
      if (have) bp = (uint64_t)(bp0 + (int64_t)rel);
      const uint64_t absbit = skip_bits + bp;
      const uint32_t cw0 = (uint32_t)(absbit >> 5);  // first dword (relative to R.base) held in s_in
      __syncthreads();
      const uint64_t first = ((uint64_t)cw0 + lane) * 4;
      const uint64_t end = (uint64_t)r_skip + r_nbytes;
      uint32_t x = 0;
      if (first < end) {
        x = r_base[cw0 + lane];
        if (first + 4 > end) x &= (1u << (8 * (uint32_t)(end - first))) - 1u;  // zeros past the end
      }
      s_in[lane] = x;
      __syncthreads();
      rel = (uint32_t)absbit & 31;
      bp0 = (int64_t)bp - (int64_t)rel;
      const int64_t l = (int64_t)nbits - bp0;
      // (l <= 0: a dynamic header may have run past the end of the input -- neither the reference nor this
      // kernel checks ReaderBitlenLeft inside it -- and then the first token must already report 2, LD:2428)
      lim = l <= 0 ? 0u : (l > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)l);
      have = true;
    }
    // ---- every lane decodes the token starting `lane` bits ahead
    bit = rel + lane;
    const uint32_t di = bit >> 5, sh = bit & 31;
    T = spec_decode(LT, DT, s_in[di], s_in[di + 1], s_in[di + 2], sh);
    const bool is_len = T.is_len, hard = T.hard;
    const uint32_t mlen = T.mlen, dist = T.dist, tok_bits = T.tok_bits, out_len = T.out_len;

    // ---- walk the chain of real tokens from lane 0.  acc = bits walked | output bytes << 8;
    // each token's lane receives acc as it was at its start.  A token the walk cannot pass
    // (slow / invalid / end of block) advances by 64 bits.
    const uint32_t P = hard ? 64u : (tok_bits | (out_len << 8));
    // The walk is six instructions a token, written out: the lane index lives in M0 (gfx9 wants v_writelane's lane
    // select there when the value written is a scalar register), v_writelane leaves acc in the token's lane.  (The
    // compiler's version of the same loop was eleven; with eight waves a SIMD this function is bound by the number of
    // instructions of every kind that a round issues, not by the vector ones alone.)
    uint32_t my_acc = 0xFFFFFFFFu, walk_t;
    acc = 0;
    asm volatile(
        "s_mov_b32 m0, 0\n"
        "1:\n\t"
        "v_writelane_b32 %[my], %[acc], m0\n\t"
        "v_readlane_b32 %[t], %[p], m0\n\t"
        "s_add_u32 %[acc], %[acc], %[t]\n\t"
        "s_and_b32 m0, %[acc], 0xff\n\t"
        "s_cmp_lt_u32 m0, 64\n\t"
        "s_cbranch_scc1 1b"
        : [my] "+v"(my_acc), [acc] "+s"(acc), [t] "=&s"(walk_t)
        : [p] "v"(P)
        : "m0", "scc");
    k = acc & 0xff;  // where the walk left the window (>= 64)
    is_tok = my_acc != 0xFFFFFFFFu;
    my_off = my_acc >> 8;
    my_pos = op + my_off;
    // a match that reads this round's own output (that includes every self-overlapping one), the dictionary, or
    // nothing at all (-11)
    const bool odd = is_len & ((my_pos + mlen > op + dist) | (dist > my_pos));
    // ---- the plain round: every token passed, the output fits, at most 64 bytes of it, every match source older than
    // the round.  Written in ONE pass, one lane per output byte: a token's lane leaves what its bytes are made of (the
    // literal, or the distance) at the token's output offset in LDS; the slots that were written are the offsets where
    // tokens start; a byte's lane takes the nearest one at or below its own; match bytes are loaded from `dist` back,
    // all matches of the round by the same load instruction (one memory latency per round, no per-match loop).
    // (LDS operations of one wave execute in order: no barrier between the writes and the reads.)
    // (Storing a round later, so that the loads overlap the next decode, was measured: 12.0 ms per GiB against 11.2.)
    {
      const uint32_t n_tot = acc >> 8;
      // (uniform tests, as branches: as one boolean they cost the scalar unit a select and an OR each)
      if (n_tot <= 64 && op + n_tot <= cap && rel + k <= lim && ballot64(is_tok & (hard | odd)) == 0) {
        s_tk[lane] = 0;
        if (is_tok) s_tk[my_off] = is_len ? (0x80000000u | dist) : (0x40000000u | T.e);  // (a literal: its entry, byte << 4)
        // (what a lane reads below was written by other lanes: without this the compiler forwards the lane's own 0)
        asm volatile("" ::: "memory");
        const unsigned long long below = ballot64(s_tk[lane] != 0) & le_mask;  // (never 0: a token starts at offset 0)
        const uint32_t info = s_tk[63 - __builtin_clzll(below)];
        // every store of earlier rounds must have landed before a match source among them is read
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // The round's bytes are stored a round later: the match sources are loaded here and waited for only behind
        // the next round's decode.  A CU serves the vector memory operations of its waves in order (that is why the
        // fences above cost no wait), so the pending store, issued just before this round's loads, is seen by them;
        // nothing reads the pending bytes before that (a source inside the round's own output is not a plain round).
        // (the pending bytes are moved to a register of their own first: that is where the loads are waited for, on
        // every path; waiting where pend_v is overwritten, behind the store's branch, would wait for the store as well)
        uint32_t st_v;
        asm volatile("v_mov_b32 %0, %1" : "=v"(st_v) : "v"(pend_v));
        if (lane < pend_n) out[pend_op + lane] = (uint8_t)st_v;
        pend_v = info >> 4;
        if ((lane < n_tot) & ((info >> 31) != 0)) pend_v = out[op + lane - (info & 0xFFFFu)];
        pend_n = n_tot;
        pend_op = op;
        op += n_tot;
        rel += k;
        continue;
      }
    }
    break;
    }
    if (lane < pend_n) out[pend_op + lane] = (uint8_t)pend_v;
    pend_n = 0;
    // ---- any other round.  Stop events, in the reference's order of checks for one token
    const bool is_len = T.is_len, hard = T.hard;
    const uint32_t mlen = T.mlen, dist = T.dist, tok_bits = T.tok_bits, out_len = T.out_len;
    const bool err11 = is_len & (dist > my_pos + ext);  // LD:2406-2410
    const bool full = my_pos + out_len > cap;
    const bool exh = bit + tok_bits > lim;              // LD:2428
    const unsigned long long stops = ballot64(is_tok & (hard | err11 | full | exh));
    const uint32_t first = stops ? (uint32_t)__builtin_ctzll(stops) : 64u;
    const bool commit = is_tok & (lane < first);
    const bool cm = commit & is_len;
    {
      // literals at once, matches in order, each one seeing everything stored before it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (commit & T.is_lit()) out[my_pos] = (uint8_t)T.lit();
      unsigned long long mm = ballot64(cm);
      while (mm) {
        const uint32_t j = (uint32_t)__builtin_ctzll(mm);
        mm &= mm - 1;
        const uint32_t len = __builtin_amdgcn_readlane(mlen, j);
        const uint32_t d = __builtin_amdgcn_readlane(dist, j);
        const uint32_t dst = op + __builtin_amdgcn_readlane(my_off, j);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (d >= len && d <= dst) {  // (uniform) no overlap with itself, source in the output
          for (uint32_t i = lane; i < len; i += kWave) out[dst + i] = out[dst - d + i];
        } else {
          for (uint32_t i = lane; i < len; i += kWave) {
            const uint32_t a = dst + (d >= len ? i : (i % d));
            out[dst + i] = a < d ? O.dict[O.dlen - (d - a)] : out[a - d];
          }
        }
      }
    }
    if (first == 64) {  // the whole window was consumed
      op += acc >> 8;
      rel += k;
      continue;
    }
    // ---- the stopping token
    const uint32_t fl = (T.slow() ? 1u : 0u) | (T.bad() ? 2u : 0u) | (err11 ? 4u : 0u) | (full ? 8u : 0u) |
                        (exh ? 16u : 0u) | (T.is_eob() ? 32u : 0u);
    const uint32_t f = __builtin_amdgcn_readlane(fl, first);
    op += __builtin_amdgcn_readlane(my_off, first);
    O.opos = op;
    rel += first;
    bp = (uint64_t)(bp0 + (int64_t)rel);
    if (f & 1) {  // unresolved by the tables: wave-uniform decode of this one token
      br_seek_bits(R, bp, lane);
      bool eob = false;
      const int st = (int)uni((uint32_t)decode_one(LT, DT, 0, 0, R, O, ext, lane, &eob));
      if (st != 0) return st;
      if (uni(eob ? 1u : 0u)) return 0;
      bp = uni64(R.consumed);
      op = uni((uint32_t)O.opos);
      have = false;
      rel = 0xFFFFFFFFu;
      continue;
    }
    R.consumed = bp;
    if (f & 2) return -10;
    if (f & 4) return -11;
    if (f & 8) return kStatusOutputFull;
    bp += __builtin_amdgcn_readlane(tok_bits, first);
    R.consumed = bp;
    if (f & 16) return 2;
    // end of block: hand the position back to the wave-uniform reader
    br_seek_bits(R, bp, lane);
    return 0;
  }
}

// ------------------------------------------------------------------ the block body for few streams
// DecodeUntilEndOfBlock (LD:2353-2445) by a workgroup of W wavefronts for ONE stream (a single string is
// what every existing caller of the reference passes, and one wavefront per stream made a 128 KiB string slower than
// the reference under LuaJIT).  A round covers W x 64 bit offsets: thread t decodes, speculatively, the token that
// would start t bits after the current position (decode_block's code).  Which of these tokens are real is a walk
// next = k + token_bits from the round's first bit.  decode_block makes that walk with the scalar unit, one dependent
// v_readlane + add per token: some 130 cycles a token for a wavefront that runs alone, so widening the window alone
// gained nothing (first version: 2.8 us per round of 256 bits, 5 us per 512).  Here the walk is made by pointer
// jumping instead: every wavefront treats each of its 64 offsets as a possible entry point and doubles
// (next, tokens visited, bytes produced) with ds_bpermute until every path has left the row (log2 of the tokens in
// 64 bits: three or four steps), publishes exit and byte count per entry point, and the rows are then chained by W
// dependent LDS reads.  The entry point of a row selects its 64-bit mask of real tokens; output offsets are a DPP
// prefix sum over it.  The round's output is written in one pass, one thread per byte, as in decode_block; a match
// that reads the round's own output (common once a round produces 100-200 bytes) or the dictionary is resolved
// through an LDS image of the round's bytes, by chasing source offsets down to a byte that is a literal or older
// output; only rounds that produce more than W x 64 bytes take the in-order path.  All wavefronts run the same
// wave-uniform control flow on the same values (each keeps its own copy of the reader and of the positions), so they
// take the same branches and meet at the same barriers.  Status codes and their order of checks are decode_block's
// (Q11); the stopping token is resolved the same way.
constexpr uint32_t kWideWin = 256;  // dwords of input staged per reload (8192 bits)

template <int W>
struct WideShared {
  uint32_t in[kWideWin + 4];
  uint32_t ex[W * kWave];    // per bit offset, as an entry point of its row: exit position | output bytes << kAccBits
  uint32_t tk[W * kWave];    // per output offset: what the token starting there is made of
  uint32_t S[2][W * 2 + 2];  // bit j: a token's output starts j bytes into the round's output (two rounds' worth)
  uint16_t ob[W * kWave];    // the round's output bytes as far as they are known (| 0x100), for matches that read them
  uint16_t sm[W * kWave];    // ... and, for a byte not yet known, the offset of the byte it copies
  uint32_t stop[W];          // first stop event among each wavefront's 64: thread << 21 | its output offset; ~0 none
  uint32_t stopf[W];         // flags << 8 | token bits of that token
  uint32_t odd[W];           // a match of this wavefront's reads the round's own output or the dictionary
  uint32_t cm[W * 2];        // in-order path: the committed matches of each wavefront's 64 offsets
  uint32_t acc[W * kWave];   // in-order path: per bit offset the token's output offset
};

// v_readfirstlane as an unsigned value (the builtin returns int: widened or shifted it would sign-extend)
__device__ inline uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); }
// packed (position | bytes << kAccBits): positions stay below 2 x 64 + 48, kHardJ marks a token the walk cannot pass
constexpr uint32_t kAccBits = 11, kAccMask = (1u << kAccBits) - 1u, kHardJ = 1024;

__device__ inline void lds_barrier() {  // LDS only: stores and loads of the output in flight are not waited for
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef LDH_WIDE_STAMPS  // diagnostic builds only: shader cycles per phase of a round (workgroup 0, wavefront 0)
__device__ unsigned long long g_wide_stamps[16];
#define LDH_STAMP(k_)                                              \
  {                                                                \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    st_acc[k_] += now_ - st_t;                                     \
    st_t = now_;                                                   \
  }
#else
#define LDH_STAMP(k_)
#endif

template <int W>
__device__ int decode_block_wide(const LitTab &LT, const DistTab &DT, BitReader &R, OutState &O, uint32_t tid,
                                 WideShared<W> &Sh) {
  constexpr uint32_t NT = W * kWave;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  uint32_t ext = 0;
  if (O.dict && !O.dict_loaded) {  // LD:2367-2377
    ext = O.dlen;
    O.dict_loaded = true;
  }
  // (The called copy gets its state through references: the compiler cannot tell that what it loads from them is
  // wave-uniform and would keep positions, pointers and the loop's conditions in vector registers.  v_readfirstlane
  // says so; in the inlined copy, where they are scalar already, it folds away.)
  const uint32_t r_nbytes = uni(R.nbytes), r_skip = uni(R.skip);
  const uint32_t *const r_base = (const uint32_t *)uni64((uint64_t)(uintptr_t)R.base);
  const uint64_t nbits = (uint64_t)r_nbytes * 8;
  const uint64_t skip_bits = (uint64_t)r_skip * 8;
  const uint32_t cap = uni((uint32_t)O.cap);
  // (a global, not a generic, pointer: loads and stores take the scalar base and a 32-bit lane offset as they are,
  // without a 64-bit vector add per access, and do not count as LDS operations in the wait counters)
  typedef __attribute__((address_space(1))) uint8_t *gbytes_t;
  const gbytes_t out = (gbytes_t)uni64((uint64_t)(uintptr_t)O.out);
  uint32_t op = uni((uint32_t)O.opos);  // output position (uniform)
  uint64_t bp = uni64(R.consumed);      // bits consumed so far (uniform); current only while !have
  int64_t bp0 = 0;                 // bits consumed = bp0 + rel while the window is valid
  uint32_t rel = 0;                // bit offset of the current position in the window
  uint32_t lim = 0;                // the input ends `lim` bits after the window's bit 0
  bool have = false;
  uint32_t round = 0;
  if (tid < 2 * (W * 2 + 2)) (&Sh.S[0][0])[tid] = 0;
#ifdef LDH_WIDE_STAMPS
  unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
  struct StampOut {
    unsigned long long *a;
    uint32_t tid;
    uint32_t *rounds;
    __device__ ~StampOut() {
      if (tid == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 10; i++) atomicAdd(&g_wide_stamps[i], a[i]);
        atomicAdd(&g_wide_stamps[15], (unsigned long long)*rounds);
      }
    }
  } stamp_out{st_acc, tid, &round};
#endif
  for (;; round++) {
    LDH_STAMP(9)
    if (!have || rel + NT + 64 > kWideWin * 32) {
      if (have) bp = (uint64_t)(bp0 + (int64_t)rel);
      const uint64_t absbit = skip_bits + bp;
      const uint32_t cw0 = (uint32_t)(absbit >> 5);  // first dword (relative to R.base) held in Sh.in
      lds_barrier();
      const uint64_t end = (uint64_t)r_skip + r_nbytes;
      for (uint32_t i = tid; i < kWideWin + 4; i += NT) {
        const uint64_t first = ((uint64_t)cw0 + i) * 4;
        uint32_t x = 0;
        if (first < end) {
          x = r_base[cw0 + i];
          if (first + 4 > end) x &= (1u << (8 * (uint32_t)(end - first))) - 1u;  // zeros past the end
        }
        Sh.in[i] = x;
      }
      lds_barrier();
      rel = (uint32_t)absbit & 31;
      bp0 = (int64_t)bp - (int64_t)rel;
      const int64_t l = (int64_t)nbits - bp0;
      lim = l <= 0 ? 0u : (l > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)l);  // (l <= 0: see decode_block)
      have = true;
    }
    uint32_t *const S = Sh.S[round & 1], *const S_next = Sh.S[(round & 1) ^ 1];
    LDH_STAMP(0)
    // ---- every thread decodes the token starting `tid` bits ahead (decode_block's code)
    const uint32_t bit = rel + tid;
    const uint32_t di = bit >> 5, sh = bit & 31;
    const SpecTok T = spec_decode(LT, DT, Sh.in[di], Sh.in[di + 1], Sh.in[di + 2], sh);
    const bool is_len = T.is_len, hard = T.hard;
    const uint32_t mlen = T.mlen, dist = T.dist, tok_bits = T.tok_bits, out_len = T.out_len;
    // ---- pointer jumping inside the wavefront's row of 64 offsets: X = where the path from this offset stands
    // (row coordinates; 64 and more: it has left the row; kHardJ: it ended at a token that cannot be passed) | the
    // bytes its tokens produce; V = the offsets it visited
    uint32_t X = (hard ? kHardJ : lane + tok_bits) | (out_len << kAccBits);
    asm volatile("" : "+v"(X));
    LDH_STAMP(1)
    uint32_t Vlo = lane < 32 ? 1u << lane : 0u, Vhi = lane >= 32 ? 1u << (lane - 32) : 0u;
    while (ballot64((X & kAccMask) < kWave)) {  // (uniform)
      const uint32_t j = X & kAccMask;
      const bool go = j < kWave;
      const int addr = (int)((go ? j : lane) << 2);
      const uint32_t Xs = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)X);
      const uint32_t Ls = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)Vlo);
      const uint32_t Hs = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)Vhi);
      if (go) {
        Vlo |= Ls;
        Vhi |= Hs;
        X = (Xs & kAccMask) | ((X >> kAccBits) + (Xs >> kAccBits)) << kAccBits;
      }
    }
    LDH_STAMP(2)
    Sh.ex[tid] = X;
    lds_barrier();
    LDH_STAMP(3)
    // ---- chain the rows: entry point and output offset of every row, this wavefront's among them.  Every wavefront
    // fetches all W rows of (exit | bytes) at once, one row per register, and then hops from row to row with
    // v_readlane (the hops through LDS, one dependent read each, were a third of a round)
    uint32_t exr[W];
#pragma unroll
    for (int r = 0; r < W; r++) exr[r] = Sh.ex[r * kWave + lane];
    uint32_t ent = 0, base = 0, my_ent = 0, my_base = 0;
    bool alive = true, my_alive = false;
#pragma unroll
    for (int r = 0; r < W; r++) {
      if ((uint32_t)r == wave) {
        my_ent = ent;
        my_base = base;
        my_alive = alive;
      }
      // (selects, no branches: `ent` stays a valid lane when the walk has ended)
      const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)exr[r], (int)ent);
      const uint32_t j = x & kAccMask;
      const bool on = alive & (j < kHardJ);
      base += alive ? x >> kAccBits : 0u;
      ent = on ? j - kWave : ent;
      alive = on;
    }
    const uint32_t total_out = base, total_bits = NT + ent;  // (total_bits: meaningful while alive)
    LDH_STAMP(4)
    // the real tokens of this row, their output offsets
    const uint32_t mlo = my_alive ? (uint32_t)__builtin_amdgcn_readlane((int)Vlo, (int)my_ent) : 0u;
    const uint32_t mhi = my_alive ? (uint32_t)__builtin_amdgcn_readlane((int)Vhi, (int)my_ent) : 0u;
    const bool is_tok = (((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1u) != 0;
    const uint32_t contrib = is_tok ? out_len : 0u;
    const uint32_t my_off = my_base + wave_inclusive_sum(contrib) - contrib;
    // ---- stop events, in the reference's order of checks for one token
    const uint32_t my_pos = op + my_off;
    const bool err11 = is_len & (dist > my_pos + ext);  // LD:2406-2410
    const bool full = my_pos + out_len > cap;
    const bool exh = bit + tok_bits > lim;              // LD:2428
    const unsigned long long stops = ballot64(is_tok & (hard | err11 | full | exh));
    // a match that reads this round's own output (that includes every self-overlapping one) or the dictionary
    const bool odd = is_tok & is_len & ((my_pos + mlen > op + dist) | (dist > my_pos));
    const unsigned long long odds = ballot64(odd);
    const uint32_t fl = (T.slow() ? 1u : 0u) | (T.bad() ? 2u : 0u) | (err11 ? 4u : 0u) | (full ? 8u : 0u) | (exh ? 16u : 0u) |
                        (T.is_eob() ? 32u : 0u);
    {
      const uint32_t sl = stops ? (uint32_t)__builtin_ctzll(stops) : 0u;
      const uint32_t s_off = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)sl);
      const uint32_t s_fl = (uint32_t)__builtin_amdgcn_readlane((int)(fl << 8 | tok_bits), (int)sl);
      if (lane == 0) {
        Sh.stop[wave] = stops ? ((wave * kWave + sl) << 21) | s_off : 0xFFFFFFFFu;
        Sh.stopf[wave] = s_fl;
        Sh.odd[wave] = odds != 0;
      }
    }
    if (is_tok & (my_off < NT)) {
      Sh.tk[my_off] = is_len ? (0x80000000u | (dist - 1)) : T.lit();
      atomicOr(&S[my_off >> 5], 1u << (my_off & 31));
    }
    if (tid < W * 2 + 2) S_next[tid] = 0;
    LDH_STAMP(5)
    // every store of earlier rounds must have landed before a match source among them is read
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    lds_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    LDH_STAMP(6)
    uint32_t st_min, st_fl, any_odd;
    {
      uint32_t sv[W], fv[W], ov[W];
#pragma unroll
      for (int r = 0; r < W; r++) {
        sv[r] = Sh.stop[r];
        fv[r] = Sh.stopf[r];
        ov[r] = Sh.odd[r];
      }
      uint32_t m = sv[0], f = fv[0], o = ov[0];
#pragma unroll
      for (int r = 1; r < W; r++) {
        f = sv[r] < m ? fv[r] : f;
        m = sv[r] < m ? sv[r] : m;
        o |= ov[r];
      }
      st_min = rfl(m);
      st_fl = rfl(f);
      any_odd = rfl(o);
    }
    const uint32_t first = st_min == 0xFFFFFFFFu ? NT : st_min >> 21;
    const uint32_t n_out = first == NT ? total_out : st_min & 0x1FFFFFu;
    LDH_STAMP(7)
    if (n_out <= NT) {
      // one thread per output byte: its token is the highest bit of S at or below its offset (bit 0 is always set)
      uint32_t v = 0, src = 0;
      bool known = true;
      if (tid < n_out) {
        uint32_t w = tid >> 5;
        uint32_t m = S[w] & (0xFFFFFFFFu >> (31 - (tid & 31)));
        while (m == 0) m = S[--w];
        const uint32_t tok_off = w * 32 + 31 - (uint32_t)__clz((int)m);
        const uint32_t info = Sh.tk[tok_off];
        v = info;
        if (info >> 31) {
          const uint32_t d = (info & 0xFFFFu) + 1;
          if (!any_odd) {  // (uniform) every source is older output
            v = out[op + tid - d];
          } else {
            // byte i of a match copies byte i mod dist of the dist bytes before the match (so a run costs one hop)
            const uint32_t i = tid - tok_off;
            const uint32_t at = tok_off + (i >= d ? i % d : i);  // its source, `d` bytes before this offset of the round
            if (at >= d) {
              known = false;
              src = at - d;
            } else {
              const uint32_t back = d - at;  // bytes before the round's first
              v = back <= op ? out[op - back] : O.dict[O.dlen - (back - op)];
            }
          }
        }
      }
      if (any_odd) {  // (uniform)
        Sh.ob[tid] = known ? (uint16_t)((v & 0xFFu) | 0x100u) : (uint16_t)0;
        Sh.sm[tid] = (uint16_t)src;
        lds_barrier();
        if (!known) {  // down the chain of sources to a byte that is known: sources lie strictly below
          uint32_t r = Sh.ob[src];
          while (!(r & 0x100u)) {
            src = Sh.sm[src];
            r = Sh.ob[src];
          }
          v = r;
        }
      }
      if (tid < n_out) out[op + tid] = (uint8_t)v;
    } else {
      // more bytes than threads: in order -- literals by their threads, then every match by the whole workgroup, each
      // one seeing what was stored before it
      const bool commit = is_tok & (tid < first);
      if (commit & T.is_lit()) out[my_pos] = (uint8_t)T.lit();
      const unsigned long long cmm = ballot64(commit & is_len);
      Sh.tk[tid] = (mlen << 16) | (dist - 1);
      Sh.acc[tid] = my_off;
      if (lane == 0) {
        Sh.cm[wave * 2] = (uint32_t)cmm;
        Sh.cm[wave * 2 + 1] = (uint32_t)(cmm >> 32);
      }
      lds_barrier();
#pragma unroll 1
      for (int r = 0; r < W; r++) {
        unsigned long long mm = ((unsigned long long)rfl(Sh.cm[r * 2 + 1]) << 32) | rfl(Sh.cm[r * 2]);
        while (mm) {
          const uint32_t j = (uint32_t)r * kWave + (uint32_t)__builtin_ctzll(mm);
          mm &= mm - 1;
          const uint32_t info = rfl(Sh.tk[j]);
          const uint32_t len = info >> 16, d = (info & 0xFFFFu) + 1;
          const uint32_t dst = op + rfl(Sh.acc[j]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          lds_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (d >= len && d <= dst) {  // (uniform) no overlap with itself, source in the output
            for (uint32_t i = tid; i < len; i += NT) out[dst + i] = out[dst - d + i];
          } else {
            for (uint32_t i = tid; i < len; i += NT) {
              const uint32_t a = dst + (d >= len ? i : (i % d));
              out[dst + i] = a < d ? O.dict[O.dlen - (d - a)] : out[a - d];
            }
          }
        }
      }
      lds_barrier();  // (tk and cm are rewritten by the next round)
    }
    LDH_STAMP(8)
    if (first == NT) {  // the whole window was consumed
      op += total_out;
      rel += total_bits;
      continue;
    }
    // ---- the stopping token
    const uint32_t f = st_fl >> 8, fbits = st_fl & 0xFFu;
    op += n_out;
    O.opos = op;
    rel += first;
    bp = (uint64_t)(bp0 + (int64_t)rel);
    if (f & 1) {  // unresolved by the tables: wave-uniform decode of this one token
      br_seek_bits(R, bp, lane);
      bool eob = false;
      const int st = decode_one<W>(LT, DT, 0, 0, R, O, ext, tid, &eob);
      if (st != 0) return st;
      if (eob) return 0;
      bp = R.consumed;
      op = (uint32_t)O.opos;
      have = false;
      continue;
    }
    R.consumed = bp;
    if (f & 2) return -10;
    if (f & 4) return -11;
    if (f & 8) return kStatusOutputFull;
    bp += fbits;
    R.consumed = bp;
    if (f & 16) return 2;
    // end of block: hand the position back to the wave-uniform reader
    br_seek_bits(R, bp, lane);
    return 0;
  }
}

// The block body can be a function of its own (BODY_CALL: kept a call by a statement attribute; a noinline wrapper
// around the inlinable body sent this compiler's backend into "Illegal instruction detected" on the LDS pointers'
// address-space casts).  In round 2 the call was the faster form with the chip full of waves (10.6 against 11.0 ms per
// GiB: inlined, the round loop shared k_inflate's scalar registers with the header code and reloaded spilled ones).
// With the round as it is now the inlined body wins at every batch size (7.0 against 7.3 ms at 8192 streams, 4.6
// against 5.0 at 2048); k_inflate, the form with the call, stays selectable (LDH_INFLATE_CALL_MIN) for comparison.
// W: wavefronts per stream (1: k_inflate / k_inflate_alone; more: k_inflate_wide, whose block body is decode_block_wide;
// everything around the body is the same wave-uniform code, run by all W wavefronts alike, stores made once).
template <int W, bool BODY_CALL>
__device__ __forceinline__ void inflate_stream(const InflateArgs &A, LitTab &LT, DistTab &DT, uint8_t *lens, uint32_t *s_in,
                                               uint32_t *s_tk, WideShared<(W > 1 ? W : 2)> *Sh) {
  constexpr uint32_t NT = W * kWave;
  const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const uint64_t off = A.in_len ? (uint64_t)s * A.in_stride : A.in_off[s];
  const uint32_t n = A.in_len ? A.in_len[s] : (uint32_t)(A.in_off[s + 1] - off);
  BitReader R;
  br_init(R, A.in + off, n, lane);
  OutState O;
  O.out = A.out + (size_t)s * A.out_stride;
  // (lengths are reported as uint32; positions inside decode_block are 32-bit)
  O.cap = A.out_stride < 0xFFF00000ull ? A.out_stride : 0xFFF00000ull;
  O.opos = 0;
  O.dict = A.dict;
  O.dlen = A.dlen;
  O.dict_loaded = false;

  int status = 0;
  uint32_t expect = 0;
  int fixed_ready = 0;
  do {
    if (A.fmt == 1) {  // LD:2664-2704
      const uint32_t cmf = br_read(R, 8, lane);
      if (br_left(R) < 0) { status = 2; break; }
      if ((cmf & 15) != 8) { status = -12; break; }
      if ((cmf >> 4) > 7) { status = -13; break; }
      const uint32_t flg = br_read(R, 8, lane);
      if (br_left(R) < 0) { status = 2; break; }
      if ((cmf * 256 + flg) % 31 != 0) { status = -14; break; }
      if ((flg >> 5) & 1) {
        if (!A.dict) { status = -16; break; }
        uint32_t ad = 0;
        for (int k = 0; k < 4; k++) ad = (ad << 8) | br_read(R, 8, lane);
        if (br_left(R) < 0) { status = 2; break; }
        if (ad != A.dict_adler) { status = -17; break; }
      }
    }
    bool last = false;
    while (!last && status == 0) {  // LD:2620-2638
      last = br_read(R, 1, lane) == 1;
      const uint32_t type = br_read(R, 2, lane);
      if (type == 0) {  // LD:2450-2491
        br_align(R, lane);
        const uint32_t len = br_read(R, 16, lane);
        if (br_left(R) < 0) { status = 2; break; }
        const uint32_t comp = br_read(R, 16, lane);
        if (br_left(R) < 0) { status = 2; break; }
        if ((len & 255) + (comp & 255) != 255 || (len >> 8) + (comp >> 8) != 255) { status = -2; break; }
        if (br_left(R) < (int64_t)len * 8) { status = 2; break; }  // ReadBytes: out of input
        if (O.opos + len > O.cap) { status = kStatusOutputFull; break; }
        const uint64_t ipos = R.consumed >> 3;
        const uint8_t *src = A.in + off + ipos;
        for (uint32_t k = tid; k < len; k += NT) O.out[O.opos + k] = src[k];
        O.opos += len;
        br_seek(R, ipos + len, lane);
      } else if (type == 1) {  // LD:2496-2501
        if (fixed_ready != 1) {
          for (int i = tid; i < 288; i += NT) lens[i] = (uint8_t)fixed_lit_bits(i);
          if (tid < 32) lens[288 + tid] = 5;
          __syncthreads();
          build_table<LitTab, W>(LT, lens, 288, 9, LitTab::kBits, tid, kLutLitLen);
          build_table<DistTab, W>(DT, lens + 288, 32, 5, DistTab::kBits, tid, kLutDist);
          fixed_ready = 1;
        }
        if constexpr (W > 1) {
          status = decode_block_wide<W>(LT, DT, R, O, tid, *Sh);
        } else if (BODY_CALL) {
          [[clang::noinline]] status = decode_block<0>(LT, DT, R, O, lane, s_in, s_tk);
        } else {
          status = decode_block<1>(LT, DT, R, O, lane, s_in, s_tk);
        }
      } else if (type == 2) {  // LD:2506-2615
        fixed_ready = 0;
        const uint32_t nlen = br_read(R, 5, lane) + 257;
        const uint32_t ndist = br_read(R, 5, lane) + 1;
        const uint32_t ncode = br_read(R, 4, lane) + 4;
        if (nlen > 286 || ndist > 30) { status = -3; break; }
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        __syncthreads();
        if (tid < 19) lens[tid] = 0;
        __syncthreads();
        for (uint32_t i = 0; i < ncode; i++) {
          const uint32_t v = br_read(R, 3, lane);
          if (tid == 0) lens[order[i]] = (uint8_t)v;
        }
        __syncthreads();
        if (build_table<DistTab, W>(DT, lens, 19, 7, 7, tid, kLutPlain) != 0) { status = -4; break; }
        __syncthreads();
        for (int i = tid; i < 320; i += NT) lens[i] = 0;
        __syncthreads();
        uint32_t index = 0;
        uint32_t prev_len = 0;  // lens[index-1], tracked in a register
        while (index < nlen + ndist) {
          int symbol = decode_sym(DT, 7, R, lane);
          if (symbol < 0) { status = symbol; break; }
          if (symbol < 16) {
            if (tid == 0) lens[index] = (uint8_t)symbol;
            prev_len = (uint32_t)symbol;
            index++;
          } else {
            uint32_t bitlen = 0, rep;
            if (symbol == 16) {
              if (index == 0) { status = -5; break; }
              bitlen = prev_len;
              rep = 3 + br_read(R, 2, lane);
            } else if (symbol == 17) {
              rep = 3 + br_read(R, 3, lane);
            } else {
              rep = 11 + br_read(R, 7, lane);
            }
            if (index + rep > nlen + ndist) { status = -6; break; }
            for (uint32_t k = tid; k < rep; k += NT) lens[index + k] = (uint8_t)bitlen;
            prev_len = bitlen;
            index += rep;
          }
        }
        if (status != 0) break;
        __syncthreads();
        if (lens[256] == 0) { status = -9; break; }
        // the distance lengths follow the nlen literal/length lengths; move them apart
        const uint8_t dl = lane < ndist ? lens[nlen + lane] : 0;
        __syncthreads();
        if (tid < 32) lens[320 + tid] = dl;
        __syncthreads();
        int err = build_table<LitTab, W>(LT, lens, (int)nlen, 15, LitTab::kBits, tid, kLutLitLen);
        if (err != 0 && (err < 0 || nlen != LT.counts[0] + LT.counts[1])) { status = -7; break; }
        err = build_table<DistTab, W>(DT, lens + 320, (int)ndist, 15, DistTab::kBits, tid, kLutDist);
        if (err != 0 && (err < 0 || ndist != DT.counts[0] + DT.counts[1])) { status = -8; break; }
        if constexpr (W > 1) {
          status = decode_block_wide<W>(LT, DT, R, O, tid, *Sh);
        } else if (BODY_CALL) {
          [[clang::noinline]] status = decode_block<0>(LT, DT, R, O, lane, s_in, s_tk);
        } else {
          status = decode_block<1>(LT, DT, R, O, lane, s_in, s_tk);
        }
      } else {
        status = -1;
      }
    }
    if (status != 0) break;
    if (A.fmt == 1) {  // LD:2707-2722 (the Adler comparison itself is k_zlib_check)
      br_align(R, lane);
      for (int k = 0; k < 4; k++) expect = (expect << 8) | br_read(R, 8, lane);
      if (br_left(R) < 0) { status = 2; break; }
    }
  } while (false);

  if (tid == 0) {
    const int64_t left = br_left(R);
    A.status[s] = status;
    A.out_len[s] = (uint32_t)O.opos;
    A.bytes_left[s] = (status == 0 && left > 0) ? (uint32_t)(left / 8) : 0u;
    if (A.expect_adler) A.expect_adler[s] = expect;
  }
}

// (two plain kernels rather than one template: with the LDS arrays inside a kernel template this compiler's backend
// stopped with "Illegal instruction detected" on their address-space casts)
#define LDH_INFLATE_KERNEL(name_, call_)                                                                      \
  __global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void name_(InflateArgs A) {  \
    __shared__ LitTab LT;                                                                                     \
    __shared__ DistTab DT;                                                                                    \
    __shared__ uint8_t lens[320 + 32];                                                                        \
    __shared__ uint32_t s_in[kWave + 4];                                                                      \
    __shared__ uint32_t s_tk[kWave];                                                                          \
    inflate_stream<1, call_>(A, LT, DT, lens, s_in, s_tk, nullptr);                                           \
  }
LDH_INFLATE_KERNEL(k_inflate, true)         // the block body as a call: batches that fill the chip
LDH_INFLATE_KERNEL(k_inflate_alone, false)  // inlined: few streams
#undef LDH_INFLATE_KERNEL

// Few streams: a workgroup of W wavefronts per stream (decode_block_wide).
#define LDH_INFLATE_WIDE_KERNEL(name_, W_)                                              \
  __global__ __launch_bounds__(W_ * kWave) void name_(InflateArgs A) {                  \
    __shared__ LitTab LT;                                                               \
    __shared__ DistTab DT;                                                              \
    __shared__ uint8_t lens[320 + 32];                                                  \
    __shared__ WideShared<W_> Sh;                                                       \
    inflate_stream<W_, false>(A, LT, DT, lens, nullptr, nullptr, &Sh);                  \
  }
LDH_INFLATE_WIDE_KERNEL(k_inflate_wide2, 2)
LDH_INFLATE_WIDE_KERNEL(k_inflate_wide4, 4)
LDH_INFLATE_WIDE_KERNEL(k_inflate_wide8, 8)
#undef LDH_INFLATE_WIDE_KERNEL

// DecompressZlibInternal's final comparison (LD:2717-2722)
__global__ void k_zlib_check(uint32_t count, const uint32_t *expect, const uint32_t *actual,
                             int32_t *status) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= count) return;
  if (status[s] == 0 && expect[s] != actual[s]) status[s] = -15;
}

}  // namespace ldh
