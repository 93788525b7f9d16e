// ldh_common.h -- shared layouts of the gfx950 DEFLATE engine (device + host).
//
// Vocabulary follows the reference (LD = LibDeflate.lua): stream = one input string,
// block = one deflate block (first 65535 bytes, then 32768 each, LD:1793-1801),
// token = literal / (length,distance) pair, chain = positions sharing a trigram.
//
// Every stream is processed over its *virtual buffer* V = dictionary ++ input.
// Virtual index v = p + dlen - 1 for LibDeflate's 1-based position p (p <= 0 is the
// dictionary).  With no dictionary dlen = 0 and v = p - 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldh {

constexpr int kWave = 64;

// HIP's __ballot takes an int: a predicate that is already a lane mask (an AND of two compares)
// then goes through v_cndmask 0/1 + v_cmp to become a mask again.  The builtin takes the bool.
__device__ inline unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
constexpr uint32_t kWindow = 32768;   // LD:1365
constexpr uint32_t kFirstBlock = 65535;  // LD:1795
constexpr uint32_t kNextBlock = 32768;   // LD:1799
constexpr uint32_t kMaxMatch = 258;
constexpr int kStatusOutputFull = 1;  // LDH_STATUS_OUTPUT_FULL
constexpr uint32_t kVPad = 32;  // zero bytes kept after V (reads past the end hash as 0, LD:1325)

// LD:657-669
struct LevelCfg {
  int lazy, good, max_lazy, nice, max_chain;
};
__host__ __device__ inline LevelCfg level_cfg(int level) {
  switch (level) {
    case 1: return {0, 0, 4, 8, 4};
    case 2: return {0, 0, 5, 18, 8};
    case 3: return {0, 0, 6, 32, 32};
    case 4: return {1, 4, 4, 16, 16};
    case 5: return {1, 8, 16, 32, 32};
    case 6: return {1, 8, 16, 128, 128};
    case 7: return {1, 8, 32, 128, 256};
    case 8: return {1, 32, 128, 258, 1024};
    case 9: return {1, 32, 258, 258, 4096};
    default: return {0, 0, 0, 0, 0};
  }
}

// LD:1793-1808 block geometry, 1-based inclusive positions.
__host__ __device__ inline uint32_t num_blocks(uint32_t n) {
  return n <= kFirstBlock ? 1u : 1u + (n - kFirstBlock + kNextBlock - 1) / kNextBlock;
}
__host__ __device__ inline uint32_t block_start(uint32_t b) {
  return b == 0 ? 1u : kFirstBlock + 1 + (b - 1) * kNextBlock;
}
__host__ __device__ inline uint32_t block_end(uint32_t b, uint32_t n) {
  uint32_t e = b == 0 ? kFirstBlock : kFirstBlock + b * kNextBlock;
  return e < n ? e : n;
}

// Token word: literal = byte value; match = 1<<31 | (len-3)<<16 | (dist-1).
constexpr uint32_t kTokMatch = 0x80000000u;
__host__ __device__ inline uint32_t tok_match(uint32_t len, uint32_t dist) {
  return kTokMatch | ((len - 3) << 16) | (dist - 1);
}
__host__ __device__ inline uint32_t tok_len(uint32_t t) { return ((t >> 16) & 0xFF) + 3; }
__host__ __device__ inline uint32_t tok_dist(uint32_t t) { return (t & 0xFFFF) + 1; }

// RFC1951 length / distance code arithmetic (same mapping as LD:341-393, LD:1408-1433).
struct Code {
  uint32_t code, ebits, eval;
};
__host__ __device__ inline int ilog2(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return 31 - __clz((int)x);
#else
  return 31 - __builtin_clz(x);
#endif
}
__host__ __device__ inline Code length_code(uint32_t len) {
  uint32_t l = len - 3;
  if (l < 8) return {257 + l, 0, 0};
  if (len == 258) return {285, 0, 0};
  int nb = ilog2(l);  // >= 3
  uint32_t g = (l >> (nb - 2)) & 3;
  return {265u + 4u * (uint32_t)(nb - 3) + g, (uint32_t)(nb - 2), l & ((1u << (nb - 2)) - 1)};
}
__host__ __device__ inline Code dist_code(uint32_t dist) {
  if (dist <= 4) return {dist - 1, 0, 0};
  uint32_t t = dist - 1;
  int nb = ilog2(t);  // >= 2
  return {2u * (uint32_t)nb + ((t >> (nb - 1)) & 1), (uint32_t)(nb - 1), t & ((1u << (nb - 1)) - 1)};
}
__host__ __device__ inline uint32_t len_extra_bits(uint32_t lcode) {  // lcode 257..285
  return (lcode < 265 || lcode == 285) ? 0u : (lcode - 261) >> 2;
}
__host__ __device__ inline uint32_t dist_extra_bits(uint32_t dcode) {
  return dcode < 4 ? 0u : (dcode >> 1) - 1;
}
__host__ __device__ inline uint32_t fixed_lit_bits(uint32_t s) {  // LD:2828-2831
  return s <= 143 ? 8u : (s <= 255 ? 9u : (s <= 279 ? 7u : 8u));
}

// Per deflate block results of the histogram / Huffman / plan kernels.
struct BlockInfo {
  uint32_t ntok;  // tokens, EOB not included
  uint32_t lcount[286];
  uint32_t dcount[30];
  uint16_t l_code[286];
  uint16_t d_code[30];
  uint8_t l_len[286];
  uint8_t d_len[30];
  uint8_t cl_len[19];
  uint8_t cl_code[19];
  uint16_t nrle;
  uint8_t rle_sym[320];
  uint8_t rle_extra[320];
  uint8_t hlit, hdist, hclen;
  uint8_t type;  // 0 store 1 fixed 2 dynamic (chosen by k_plan)
  uint32_t dyn_bits, fix_bits;
  uint64_t bit_start;  // absolute bit offset of the block inside the stream's output
  uint64_t bit_end;
};

// One batch of streams, passed by value to every compress kernel.
struct Batch {
  const uint8_t *in;
  const uint64_t *in_off;  // count+1
  uint32_t count;
  const uint8_t *dict;
  uint32_t dlen;
  uint32_t dict_adler;
  uint32_t max_len;     // caller's bound on a stream's length
  uint32_t nvp;         // per-stream stride (elements) of the per-position arrays
  uint32_t max_blocks;  // num_blocks(max_len)
  int level, strategy, fmt;
  // workspace
  uint8_t *vbuf;        // count * (nvp + kVPad)  virtual buffers
  uint16_t *prevd;      // count * nvp   distance to the previous same-trigram position (0 none)
  uint32_t *bufA;       // count * nvp   sort scratch, then tokens
  uint32_t *bufB;       // count * nvp   sort scratch, then per-position best match (lazy) / rank of a position in the sorted array (greedy)
  uint32_t *bufC;       // count * 4*nvp greedy levels only: the sorted array as {position, its first 12 bytes} entries
  uint32_t sort_lds;        // bytes of dynamic LDS k_sort_links<true> is launched with (0 for <false>)
  uint32_t need_lz;     // 0: level 0 / huffman_only, k_sort_links only materialises V
  uint32_t want_rank;   // k_sort_links leaves sorted array in bufC and its inverse in bufB (greedy levels)
  BlockInfo *blocks;    // count * max_blocks
  uint32_t *adler;      // count
  // results
  uint8_t *out;
  uint64_t out_stride;
  uint32_t *out_len;
  uint8_t *pad_bits;
  uint32_t *too_long;   // device counter: streams longer than max_len (reported by ldh_sync)
};

__device__ inline uint32_t stream_len(const Batch &B, uint32_t s) {
  uint64_t n = B.in_off[s + 1] - B.in_off[s];
  // (a stream longer than the caller's bound is rejected: k_plan counts it in *too_long, leaves out_len 0,
  // and ldh_sync() fails with LDH_E_TOO_LONG; the clamp only keeps the other kernels inside their buffers)
  return n > B.max_len ? B.max_len : (uint32_t)n;
}

// x + (x moved along by a DPP control; lanes without a source lane add 0)
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_add(uint32_t x) {
  return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
}
__device__ inline uint32_t wave_inclusive_sum(uint32_t x) {
  x = dpp_add<0x111, 0xF>(x);  // row_shr:1
  x = dpp_add<0x112, 0xF>(x);  // row_shr:2
  x = dpp_add<0x114, 0xF>(x);  // row_shr:4
  x = dpp_add<0x118, 0xF>(x);  // row_shr:8   -> sums within rows of 16
  x = dpp_add<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3
  x = dpp_add<0x143, 0xC>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}

// max(x, x moved along by a DPP control); lanes without a source lane keep x (values are unsigned)
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_max(uint32_t x) {
  const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
  return y > x ? y : x;
}
__device__ inline uint32_t wave_inclusive_max(uint32_t x) {
  x = dpp_max<0x111, 0xF>(x);  // row_shr:1
  x = dpp_max<0x112, 0xF>(x);  // row_shr:2
  x = dpp_max<0x114, 0xF>(x);  // row_shr:4
  x = dpp_max<0x118, 0xF>(x);  // row_shr:8   -> maxima within rows of 16
  x = dpp_max<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3
  x = dpp_max<0x143, 0xC>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}

// the value of the lane to the left / right (lane 0 / lane 63 get 0): DPP wave shifts
__device__ inline uint32_t wave_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, false); }
__device__ inline uint32_t wave_shl1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, false); }

}  // namespace ldh
