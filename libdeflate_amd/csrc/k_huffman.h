// k_huffman.h -- north-star parts (2) histograms, (3) Huffman build + header RLE + block
// choice, (4) bit packing.
//
//   k_histogram  LD:1436-1440,1470,1481   LDS-atomic literal/length and distance counts
//   k_huffman    LD:968-1085, 1099-1174, 1491-1567, 1646-1670: one wavefront per block
//   k_plan       LD:1718-1727, 1867-1894, 1963-1965, 1970-2024: serial over a stream's
//                blocks (stored-block padding depends on the absolute bit position)
//   k_bitpack    LD:736-832, 1571-1640, 1675-1710, 1732-1750: prefix scan of code lengths,
//                bits OR-ed into an LDS image, whole words stored coalesced
#pragma once
#include "ldh_common.h"

namespace ldh {

// ------------------------------------------------------------------ k_histogram
__global__ __launch_bounds__(256) void k_histogram(Batch B) {
  __shared__ uint32_t h[320];
  const uint32_t s = blockIdx.x / B.max_blocks, b = blockIdx.x % B.max_blocks;
  const uint32_t n = stream_len(B, s);
  if (b >= num_blocks(n)) return;
  BlockInfo *bi = &B.blocks[(size_t)s * B.max_blocks + b];
  const uint32_t *tok = B.bufA + (size_t)s * B.nvp + (block_start(b) - 1);
  const uint32_t ntok = bi->ntok;
  for (uint32_t i = threadIdx.x; i < 320; i += 256) h[i] = 0;
  __syncthreads();
  // (eight loads in flight per thread: one at a time, each waited for, was the kernel's time)
  for (uint32_t i0 = threadIdx.x; i0 < ntok; i0 += 8 * 256) {
    uint32_t tv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t i = i0 + k * 256;
      tv[k] = tok[i < ntok ? i : ntok - 1];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t t = tv[k];
      if (i0 + k * 256 >= ntok) break;
      if (t & kTokMatch) {
        atomicAdd(&h[length_code(tok_len(t)).code], 1u);
        atomicAdd(&h[286 + dist_code(tok_dist(t)).code], 1u);
      } else {
        atomicAdd(&h[t], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 286; i += 256) bi->lcount[i] = h[i] + (i == 256 ? 1u : 0u);  // EOB, LD:1481
  for (uint32_t i = threadIdx.x; i < 30; i += 256) bi->dcount[i] = h[286 + i];
}

// ------------------------------------------------------------------ k_huffman
// (4.9 KB with the kernel's own arrays, so that 32 wavefronts share a CU: the heap is a chain of
// dependent LDS accesses on one lane, and what the CU gets done is the number of heaps in flight.
// Tree node i < nleaf is the leaf leafs[i]; the codes reuse the heap's slots once the tree stands.)
struct HuffScratch {
  int16_t leafs[288];
  uint32_t hp[292];  // heap slots: weight << 10 | node; afterwards the codes (uint16_t[288])
  uint32_t cnt[288];  // counts of the alphabet being built
  uint8_t len[288];
  int bl_count[17];
  int next_code[17];
  int16_t parent[576];
  int nnodes;
};

// The binary heap of LD:842-903 with its exact comparisons (children: the right one only if strictly
// lighter; moves only while strictly lighter / heavier), so that ties fall as in the reference (Q7).
// A heap slot holds weight << 10 | node, so a comparison costs one LDS read instead of two dependent
// ones, and the sifted element is written once, where it ends up, instead of at every level (same
// final array).  Weights are < 2^22 (a block has at most 65536 symbols).
// LD:863-903
__device__ inline uint32_t hs_pop(HuffScratch &S, int heap_size) {
  const uint32_t top = S.hp[1];
  const uint32_t e = S.hp[heap_size];
  const uint32_t value = e >> 10;
  heap_size--;
  int pos = 1, l = 2;
  while (l <= heap_size) {
    uint32_t c = S.hp[l];
    int cp = l;
    if (l + 1 <= heap_size) {
      const uint32_t rc = S.hp[l + 1];
      if ((rc >> 10) < (c >> 10)) {
        c = rc;
        cp = l + 1;
      }
    }
    if ((c >> 10) < value) {
      S.hp[pos] = c;
      pos = cp;
      l = pos * 2;
    } else {
      break;
    }
  }
  S.hp[pos] = e;
  return top;
}
// LD:842-856
__device__ inline void hs_push(HuffScratch &S, uint32_t e, int heap_size) {
  const uint32_t value = e >> 10;
  int pos = heap_size + 1, parent = pos / 2;
  while (parent >= 1) {
    const uint32_t pe = S.hp[parent];
    if ((pe >> 10) <= value) break;
    S.hp[pos] = pe;
    pos = parent;
    parent /= 2;
  }
  S.hp[pos] = e;
}

// GetHuffmanBitlenAndCode (LD:968-1085) over S.cnt[0..nsym); results in S.len and the codes (uint16_t[nsym] over S.hp).
// Called by the whole wavefront; the leaf sort is done by all lanes, the heap by lane 0.
// Returns max_non_zero_bitlen_symbol (-1 if no symbol).
__device__ int huff_build(HuffScratch &S, int nsym, int max_bitlen, uint32_t lane) {
  uint16_t *const code = (uint16_t *)S.hp;  // (zeroed and written after the heap is done with its slots)
  for (int i = lane; i < nsym; i += kWave) S.len[i] = 0;
  // rank of every used symbol in (count, symbol) order (LD:953-955, 1002).  The used symbols are first gathered
  // (count << 9 | symbol: distinct keys, the order wanted is their numeric order) into the heap's slots, which are free
  // until the heap is built, so that a symbol is compared with the used ones only: a 4 KiB block uses some 80 of the
  // 286 literal/length symbols, and comparing every symbol with every symbol was a quarter of the kernel.
  int nleaf = 0;
  for (int base = 0; base < nsym; base += kWave) {
    const int i = base + (int)lane;
    const uint32_t ci = i < nsym ? S.cnt[i] : 0;
    const unsigned long long used = ballot64(ci != 0);
    if (ci) S.hp[nleaf + __popcll(used & ((1ull << lane) - 1ull))] = (ci << 9) | (uint32_t)i;
    nleaf += __popcll(used);
  }
  __syncthreads();
  for (int base = 0; base < nleaf; base += kWave) {
    const int i = base + (int)lane;
    const uint32_t ki = i < nleaf ? S.hp[i] : 0;
    int rank = 0;
    for (int j = 0; j < nleaf; j++) rank += S.hp[j] < ki;
    if (i < nleaf) S.leafs[rank] = (int16_t)(ki & 511u);
  }
  __syncthreads();
  int max_nz = -1;
  if (lane == 0) {
    if (nleaf == 1) {  // LD:992-1000
      const int sy = S.leafs[0];
      S.len[sy] = 1;  // (its code, 0, is set with all the others below)
      max_nz = sy;
    } else if (nleaf >= 2) {
      for (int i = 0; i < nleaf; i++) {
        const int sy = S.leafs[i];
        S.hp[i + 1] = (S.cnt[sy] << 10) | (uint32_t)i;
      }
      int heap_size = nleaf, nnodes = nleaf;
      while (heap_size > 1) {  // LD:1006-1015
        const uint32_t l = hs_pop(S, heap_size);
        heap_size--;
        const uint32_t r = hs_pop(S, heap_size);
        heap_size--;
        S.parent[l & 1023] = (int16_t)nnodes;
        S.parent[r & 1023] = (int16_t)nnodes;
        hs_push(S, (((l >> 10) + (r >> 10)) << 10) | (uint32_t)nnodes, heap_size);
        nnodes++;
        heap_size++;
      }
      S.parent[S.hp[1] & 1023] = -1;
      S.nnodes = nnodes;
    }
  }
  __syncthreads();
  for (int i = lane; i < nsym; i += kWave) code[i] = 0;
  if (nleaf >= 2) {
    // Depths (LD:1021-1053).  The reference walks the tree breadth first, but what it computes does
    // not depend on the order: a node's depth, `overflow` = number of nodes (internal ones too) deeper
    // than max_bitlen, leaf lengths clamped to max_bitlen, their histogram.  Every lane climbs from
    // its nodes to the root.
    const int nnodes = S.nnodes;
    if (lane <= 16) S.bl_count[lane] = 0;
    __syncthreads();
    int over = 0;
    for (int i = lane; i < nnodes; i += kWave) {
      int depth = 0;
      for (int p = S.parent[i]; p >= 0; p = S.parent[p]) depth++;
      if (depth > max_bitlen) {
        over++;
        depth = max_bitlen;
      }
      const int sy = i < nleaf ? S.leafs[i] : -1;
      if (sy >= 0) {
        S.len[sy] = (uint8_t)depth;
        atomicAdd(&S.bl_count[depth], 1);
        if (sy > max_nz) max_nz = sy;
      }
    }
    // (DPP scans, the totals read from lane 63: butterflies of __shfl_xor are trips through the LDS crossbar)
    over = (int)__builtin_amdgcn_readlane((int)wave_inclusive_sum((uint32_t)over), 63);
    max_nz = (int)__builtin_amdgcn_readlane((int)wave_inclusive_max((uint32_t)(max_nz + 1)), 63) - 1;
    __syncthreads();
    if (over > 0 && lane == 0) {  // LD:1057-1079
      int overflow = over, bitlen, index = 0;
      do {
        bitlen = max_bitlen - 1;
        while (bitlen > 0 && S.bl_count[bitlen] == 0) bitlen--;
        if (bitlen <= 0) break;
        S.bl_count[bitlen]--;
        S.bl_count[bitlen + 1] += 2;
        S.bl_count[max_bitlen]--;
        overflow -= 2;
      } while (overflow > 0);
      for (bitlen = max_bitlen; bitlen >= 1; bitlen--) {
        int k = S.bl_count[bitlen];
        while (k > 0) {
          S.len[S.leafs[index]] = (uint8_t)bitlen;
          k--;
          index++;
        }
      }
    }
    __syncthreads();
    // LD:913-948 canonical codes, bit-reversed: first code of each length, then symbols of one
    // length in increasing order (rank among the wave's lanes by ballots, running count per length)
    if (lane == 0) {
      int code = 0;
      S.next_code[0] = 0;
      for (int bl = 1; bl <= max_bitlen; bl++) {
        code = (code + S.bl_count[bl - 1]) * 2;
        S.next_code[bl] = code;
      }
    }
    __syncthreads();
    for (int base = 0; base < nsym; base += kWave) {
      const int sy = base + (int)lane;
      const uint32_t len = sy < nsym ? S.len[sy] : 0;
      const bool active = len != 0;
      unsigned long long mask = ballot64(active);
#pragma unroll
      for (int bit = 0; bit < 4; bit++) {
        const unsigned long long bal = ballot64(active && ((len >> bit) & 1));
        mask &= ((len >> bit) & 1) ? bal : ~bal;
      }
      const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (active) {
        const uint32_t c = (uint32_t)S.next_code[len] + rank;
        code[sy] = (uint16_t)(__brev(c) >> (32 - len));
      }
      __syncthreads();
      if (active && rank == 0) S.next_code[len] += (int)__popcll(mask);
      __syncthreads();
    }
  }
  __syncthreads();
  max_nz = __shfl(max_nz, 0);
  return max_nz;
}

__device__ inline int rle_extra_bitlen(int c) { return c == 16 ? 2 : (c == 17 ? 3 : 7); }

__global__ __launch_bounds__(kWave) void k_huffman(Batch B) {
  __shared__ HuffScratch S;
  __shared__ uint8_t s_llen[286], s_dlen[30];
  __shared__ uint32_t s_rlecnt[19];
  const uint32_t s = blockIdx.x / B.max_blocks, b = blockIdx.x % B.max_blocks;
  const uint32_t n = stream_len(B, s);
  if (b >= num_blocks(n)) return;
  const uint32_t lane = threadIdx.x;
  BlockInfo *bi = &B.blocks[(size_t)s * B.max_blocks + b];

  // literal/length tree (LD:1493)
  for (int i = lane; i < 286; i += kWave) S.cnt[i] = bi->lcount[i];
  __syncthreads();
  const int max_l = huff_build(S, 286, 15, lane);
  uint64_t dyn = 0, fix = 0;
  for (int i = lane; i < 286; i += kWave) {
    const uint32_t c = S.cnt[i], l = S.len[i];
    bi->l_len[i] = (uint8_t)l;
    bi->l_code[i] = ((const uint16_t *)S.hp)[i];
    s_llen[i] = (uint8_t)l;
    const uint32_t eb = i > 256 ? len_extra_bits(i) : 0;
    dyn += (uint64_t)c * (l + eb);
    fix += (uint64_t)c * (fixed_lit_bits(i) + eb);
  }
  __syncthreads();
  // distance tree (LD:1495)
  for (int i = lane; i < 30; i += kWave) S.cnt[i] = bi->dcount[i];
  __syncthreads();
  int max_d = huff_build(S, 30, 15, lane);
  for (int i = lane; i < 30; i += kWave) {
    const uint32_t c = S.cnt[i], l = S.len[i];
    bi->d_len[i] = (uint8_t)l;
    bi->d_code[i] = ((const uint16_t *)S.hp)[i];
    s_dlen[i] = (uint8_t)l;
    const uint32_t eb = dist_extra_bits(i);
    dyn += (uint64_t)c * (l + eb);
    fix += (uint64_t)c * (5 + eb);
  }
  for (int i = lane; i < 19; i += kWave) s_rlecnt[i] = 0;
  __syncthreads();

  // RunLengthEncodeHuffmanBitlen (LD:1099-1174), serial: <= 318 steps
  int nrle = 0;
  uint32_t rle_bits_extra = 0;
  if (lane == 0) {
    int prev = -1, count = 0;
    int md = max_d < 0 ? 0 : max_d;
    const int max_code = max_l + md + 1;
    for (int code = 0; code <= max_code + 1; code++) {
      const int len = code <= max_l ? (int)s_llen[code]
                                    : (code <= max_code ? (int)s_dlen[code - max_l - 1] : -2);
      if (len == prev) {
        count++;
        if (len != 0 && count == 6) {
          bi->rle_sym[nrle] = 16;
          bi->rle_extra[nrle] = 3;
          nrle++;
          s_rlecnt[16]++;
          count = 0;
        } else if (len == 0 && count == 138) {
          bi->rle_sym[nrle] = 18;
          bi->rle_extra[nrle] = 127;
          nrle++;
          s_rlecnt[18]++;
          count = 0;
        }
      } else {
        if (count == 1) {
          bi->rle_sym[nrle] = (uint8_t)prev;
          bi->rle_extra[nrle] = 0;
          nrle++;
          s_rlecnt[prev]++;
        } else if (count == 2) {
          bi->rle_sym[nrle] = (uint8_t)prev;
          bi->rle_extra[nrle] = 0;
          nrle++;
          bi->rle_sym[nrle] = (uint8_t)prev;
          bi->rle_extra[nrle] = 0;
          nrle++;
          s_rlecnt[prev] += 2;
        } else if (count >= 3) {
          const int rc = prev != 0 ? 16 : (count <= 10 ? 17 : 18);
          bi->rle_sym[nrle] = (uint8_t)rc;
          bi->rle_extra[nrle] = (uint8_t)(count <= 10 ? count - 3 : count - 11);
          nrle++;
          s_rlecnt[rc]++;
        }
        prev = len;
        if (len > 0) {
          bi->rle_sym[nrle] = (uint8_t)len;
          bi->rle_extra[nrle] = 0;
          nrle++;
          s_rlecnt[len]++;
          count = 0;
        } else {
          count = 1;
        }
      }
    }
    bi->nrle = (uint16_t)nrle;
  }
  __syncthreads();
  // code-length code (LD:1504)
  for (int i = lane; i < 19; i += kWave) S.cnt[i] = s_rlecnt[i];
  __syncthreads();
  huff_build(S, 19, 7, lane);
  if (lane < 19) {
    bi->cl_len[lane] = S.len[lane];
    bi->cl_code[lane] = (uint8_t)((const uint16_t *)S.hp)[lane];
    const uint32_t c = S.cnt[lane];
    const uint32_t eb = lane >= 16 ? (uint32_t)rle_extra_bitlen((int)lane) : 0;
    rle_bits_extra = c * (S.len[lane] + eb);
  }
  // wave reductions
  // (a block's bit counts stay below 2^32; DPP scans, the totals read from lane 63)
  dyn = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum((uint32_t)dyn), 63);
  fix = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum((uint32_t)fix), 63);
  rle_bits_extra = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(rle_bits_extra), 63);
  if (lane == 0) {
    const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 0;
    for (int i = 1; i <= 19; i++)
      if (S.len[order[i - 1]] != 0) hclen = i;
    hclen -= 4;  // LD:1513
    int hdist = max_d + 1 - 1;
    if (hdist < 0) hdist = 0;
    bi->hclen = (uint8_t)hclen;
    bi->hlit = (uint8_t)(max_l + 1 - 257);
    bi->hdist = (uint8_t)hdist;
    bi->dyn_bits = (uint32_t)(17 + 3 * (hclen + 4) + rle_bits_extra + dyn);  // LD:1526-1567
    bi->fix_bits = (uint32_t)(3 + fix);                                       // LD:1646-1670
  }
}

// ------------------------------------------------------------------ k_plan
// One lane per stream.  LD:1865-1894 block choice (ties: store, then fixed, then dynamic),
// running total_bitlen, final padding / zlib trailer sizes.  Also zeroes the output words
// that two blocks share so that k_bitpack may OR into them.
// One block's choice (LD:1867-1894) at absolute bit position `total`: its type and size in bits.
struct PlanChoice {
  uint32_t type;
  uint64_t bits;
};
__device__ inline PlanChoice plan_block(const Batch &B, uint32_t b, uint32_t n, uint64_t total, uint64_t fix, uint64_t dyn) {
  const uint32_t bs = block_start(b), be = block_end(b, n);
  const uint64_t t3 = total + 3;
  const uint64_t store = 3 + (8 - t3 % 8) % 8 + 32 + (uint64_t)(be + 1 - bs) * 8;  // LD:1718-1727
  if (B.level == 0) return {0u, store};
  uint64_t mn = store;
  if (fix < mn) mn = fix;
  if (dyn < mn) mn = dyn;
  if (B.strategy != 1 && B.strategy != 3 && store == mn) return {0u, store};
  if (B.strategy != 3 && (B.strategy == 1 || fix == mn)) return {1u, fix};
  return {2u, dyn};
}
// what k_plan leaves for a block that starts at `total`: its record, and zeroed words where k_bitpack will OR
__device__ inline void plan_commit(BlockInfo *bi, uint32_t *out32, uint64_t total, PlanChoice c) {
  bi->type = (uint8_t)c.type;
  bi->bit_start = total;
  bi->bit_end = total + c.bits;
  out32[total >> 5] = 0;
  // stored block: its header ends mid-word and raw bytes follow in the same word
  if (c.type == 0) out32[(total + 3 + (8 - (total + 3) % 8) % 8 + 32) >> 5] = 0;
}
__device__ inline void plan_finish(const Batch &B, uint32_t s, uint32_t *out32, uint64_t total) {
  out32[total >> 5] = 0;
  const uint32_t pad = (uint32_t)((8 - total % 8) % 8);
  uint64_t bytes = (total + pad) / 8;
  if (B.fmt == 1) {
    out32[(total + pad + 31) >> 5] = 0;  // trailer may end in a fresh word
    bytes += 4;
  }
  if (B.in_off[s + 1] - B.in_off[s] > (uint64_t)B.max_len) {  // not truncated silently: rejected
    atomicAdd(B.too_long, 1u);
    B.out_len[s] = 0;
    B.pad_bits[s] = 0;
    return;
  }
  B.out_len[s] = (uint32_t)bytes;
  B.pad_bits[s] = (uint8_t)(B.fmt == 1 ? 0 : pad);
}

__global__ void k_plan(Batch B) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.count) return;
  const uint32_t n = stream_len(B, s);
  const uint32_t nblk = num_blocks(n);
  uint32_t *out32 = (uint32_t *)(B.out + (size_t)s * B.out_stride);
  uint64_t total = 0;
  if (B.fmt == 1) total = B.dict ? 48 : 16;  // CMF, FLG, DICTID (LD:1976-2001)
  out32[0] = 0;  // the zlib header rides with block 0 and is OR-ed in as well
  for (uint32_t b = 0; b < nblk; b++) {
    BlockInfo *bi = &B.blocks[(size_t)s * B.max_blocks + b];
    const PlanChoice c = plan_block(B, b, n, total, B.level ? bi->fix_bits : 0, B.level ? bi->dyn_bits : 0);
    plan_commit(bi, out32, total, c);
    total += c.bits;
  }
  plan_finish(B, s, out32, total);
}

// The same for streams of many blocks (one long stream has thousands): one wavefront per stream.  A block's
// size depends on where it starts only through the padding of a stored block, i.e. on the start's bit position
// mod 8 (which can also tip the choice of type): a run of blocks is a map from the entry phase to the bits it
// adds.  Every lane takes a run, evaluates it for the eight phases, the 64 maps are chained from lane 0 (64
// table look-ups), and every lane walks its run again from its true start.  2 * blocks / 64 + 64 steps instead
// of `blocks` (one 256 MiB stream: 3.1 -> ms of the 13 its compression took).
__global__ __launch_bounds__(kWave) void k_plan_wave(Batch B) {
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  const uint32_t n = stream_len(B, s);
  const uint32_t nblk = num_blocks(n);
  uint32_t *out32 = (uint32_t *)(B.out + (size_t)s * B.out_stride);
  uint64_t start = 0;
  if (B.fmt == 1) start = B.dict ? 48 : 16;
  if (lane == 0) out32[0] = 0;
  const uint32_t per = (nblk + kWave - 1) / kWave;
  const uint32_t b0 = lane * per < nblk ? lane * per : nblk, b1 = b0 + per < nblk ? b0 + per : nblk;
  BlockInfo *bi0 = &B.blocks[(size_t)s * B.max_blocks];
  uint64_t add[8];  // bits my run adds when it is entered at phase m
#pragma unroll
  for (int m = 0; m < 8; m++) add[m] = 0;
  for (uint32_t b = b0; b < b1; b++) {
    const uint64_t fix = B.level ? bi0[b].fix_bits : 0, dyn = B.level ? bi0[b].dyn_bits : 0;
#pragma unroll
    for (int m = 0; m < 8; m++) add[m] += plan_block(B, b, n, (uint64_t)m + add[m], fix, dyn).bits;
  }
  // chain the runs: lane l starts where lanes 0..l-1 end
  uint64_t total = start, mine = 0;
  for (uint32_t l = 0; l < kWave; l++) {
    if (lane == l) mine = total;
    uint64_t a = 0;
#pragma unroll
    for (int m = 0; m < 8; m++)
      if ((total & 7) == (uint64_t)m)
        a = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(add[m] >> 32), l) << 32) | __builtin_amdgcn_readlane((uint32_t)add[m], l);
    total += a;
  }
  uint64_t t = mine;
  for (uint32_t b = b0; b < b1; b++) {
    const PlanChoice c = plan_block(B, b, n, t, B.level ? bi0[b].fix_bits : 0, B.level ? bi0[b].dyn_bits : 0);
    plan_commit(&bi0[b], out32, t, c);
    t += c.bits;
  }
  if (lane == 0) plan_finish(B, s, out32, total);
}

// ------------------------------------------------------------------ k_bitpack
constexpr int kPackThreads = 256;
constexpr int kPackItems = 4;                               // symbols per thread per round
constexpr int kPackRound = kPackThreads * kPackItems;       // 1024 symbols, <= 48 bits each
constexpr int kPackWords = (kPackRound * 48) / 32 + 8;      // LDS image of one round

struct Packer {
  uint32_t *img;        // LDS image, kPackWords words
  uint32_t *scan;       // LDS scratch for the block scan, kPackThreads/64 + 1 words
  uint32_t *out32;      // stream's output as words
  uint64_t bitpos;      // absolute bit position of the next symbol (uniform)
  uint64_t first_word;  // word index img[0] stands for
  uint64_t block_first_word;  // first word of this deflate block (shared with predecessor)
  bool block_first_partial;
};

// OR `nbits` (<= 57) bits of `val` into the LDS image at absolute bit position `pos`.
__device__ inline void img_or(const Packer &P, uint64_t pos, uint64_t val, uint32_t nbits) {
  if (!nbits) return;
  const uint32_t rel = (uint32_t)(pos - P.first_word * 32);
  const uint32_t w = rel >> 5, sh = rel & 31;
  atomicOr(&P.img[w], (uint32_t)(val << sh));
  if (sh + nbits > 32) {
    const uint64_t hi = val >> (32 - sh);
    atomicOr(&P.img[w + 1], (uint32_t)hi);
    if (sh + nbits > 64) atomicOr(&P.img[w + 2], (uint32_t)(hi >> 32));
  }
}

// All threads call this once per round with their symbols' (value, nbits); the round's
// bits are appended at P.bitpos.  Completed words go to global memory.
__device__ inline void pack_round(Packer &P, const uint64_t *val, const uint32_t *nb, bool last_round) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < kPackItems; k++) mine += nb[k];
  const uint32_t inc = wave_inclusive_sum(mine);
  if (lane == 63) P.scan[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < kPackThreads / 64; w++) {
    const uint32_t t = P.scan[w];
    if (w < wave) base += t;
    tot += t;
  }
  uint64_t pos = P.bitpos + base + inc - mine;
#pragma unroll
  for (int k = 0; k < kPackItems; k++) {
    img_or(P, pos, val[k], nb[k]);
    pos += nb[k];
  }
  __syncthreads();
  const uint64_t end = P.bitpos + tot;
  // words [first_word, end/32) are complete; on the last round of the block also flush the tail
  const uint64_t full_end = end >> 5;
  const uint32_t nfull = (uint32_t)(full_end - P.first_word);
  const uint32_t tail = (uint32_t)(end & 31);
  for (uint32_t i = tid; i < nfull; i += kPackThreads) {
    const uint64_t gw = P.first_word + i;
    if (gw == P.block_first_word && P.block_first_partial)
      atomicOr(&P.out32[gw], P.img[i]);
    else
      P.out32[gw] = P.img[i];
  }
  uint32_t carry = (tail || last_round) ? P.img[nfull] : 0;
  __syncthreads();
  if (last_round) {
    if (tid == 0 && tail) atomicOr(&P.out32[full_end], carry);
  }
  // reset image: word 0 = carry (partial word), rest zero
  for (uint32_t i = tid; i < (uint32_t)kPackWords; i += kPackThreads) P.img[i] = (i == 0 && !last_round) ? carry : 0;
  P.first_word = full_end;
  P.bitpos = end;
  __syncthreads();
}

__global__ __launch_bounds__(kPackThreads) void k_bitpack(Batch B) {
  __shared__ uint32_t s_img[kPackWords];
  __shared__ uint32_t s_scan[kPackThreads / 64 + 1];
  __shared__ uint32_t s_lc[286], s_dc[30];
  const uint32_t s = blockIdx.x / B.max_blocks, b = blockIdx.x % B.max_blocks;
  const uint32_t n = stream_len(B, s);
  const uint32_t nblk = num_blocks(n);
  if (b >= nblk) return;
  const uint32_t tid = threadIdx.x;
  const BlockInfo *bi = &B.blocks[(size_t)s * B.max_blocks + b];
  const uint32_t bs = block_start(b), be = block_end(b, n);
  const bool last = (b + 1 == nblk);
  const uint32_t type = bi->type;
  uint8_t *out = B.out + (size_t)s * B.out_stride;

  Packer P;
  P.img = s_img;
  P.scan = s_scan;
  P.out32 = (uint32_t *)out;
  uint64_t start = bi->bit_start;
  // stream prologue (zlib header) rides with block 0
  const bool zhdr = (b == 0 && B.fmt == 1);
  if (zhdr) start = 0;
  P.bitpos = start;
  P.first_word = start >> 5;
  P.block_first_word = P.first_word;
  P.block_first_partial = true;  // the block's first word may be shared; k_plan zeroed it
  for (uint32_t i = tid; i < (uint32_t)kPackWords; i += kPackThreads) s_img[i] = 0;
  // the block's codes (code | length << 16) go to LDS once: per token they were two to four
  // gathers from BlockInfo in global memory
  if (type == 2) {
    for (uint32_t i = tid; i < 286; i += kPackThreads) s_lc[i] = (uint32_t)bi->l_code[i] | ((uint32_t)bi->l_len[i] << 16);
    if (tid < 30) s_dc[tid] = (uint32_t)bi->d_code[tid] | ((uint32_t)bi->d_len[tid] << 16);
  }
  __syncthreads();

  uint64_t val[kPackItems];
  uint32_t nb[kPackItems];

  // ---- header symbols (few): produced by thread-serial lists, packed in rounds
  // sequence: [zlib header bytes] BFINAL/BTYPE [store: pad, LEN, NLEN | dynamic: HLIT.. cl lens, rle codes]
  {
    // number of header items
    uint32_t nh_z = zhdr ? (B.dict ? 6u : 2u) : 0u;
    uint32_t nh = nh_z + 1;
    if (type == 0) nh += 2;  // pad+LEN, NLEN
    if (type == 2) nh += 1 + (bi->hclen + 4u) + bi->nrle;
    for (uint32_t r0 = 0; r0 < nh; r0 += kPackRound) {
#pragma unroll
      for (int k = 0; k < kPackItems; k++) {
        const uint32_t i = r0 + tid * kPackItems + k;
        uint64_t v = 0;
        uint32_t l = 0;
        if (i < nh) {
          if (i < nh_z) {
            const uint32_t cmf = 0x78;
            uint32_t flg = 2 * 64 + (B.dict ? 32 : 0);
            flg += 31 - (cmf * 256 + flg) % 31;
            if (i == 0) v = cmf;
            else if (i == 1) v = flg;
            else v = (B.dict_adler >> (8 * (5 - i))) & 0xFF;  // DICTID big-endian (LD:1988-2001)
            l = 8;
          } else {
            const uint32_t j = i - nh_z;
            if (j == 0) {
              v = (last ? 1u : 0u) | (type << 1);
              l = 3;
            } else if (type == 0) {
              const uint32_t size = be + 1 - bs;
              if (j == 1) {  // LD:1737-1743 pad with ones, then LEN
                const uint32_t pad = (uint32_t)((8 - (bi->bit_start + 3) % 8) % 8);
                v = ((1u << pad) - 1u) | ((uint64_t)size << pad);
                l = pad + 16;
              } else {
                v = (~size) & 0xFFFF;
                l = 16;
              }
            } else {  // dynamic header LD:1585-1604
              if (j == 1) {
                v = (uint32_t)bi->hlit | ((uint32_t)bi->hdist << 5) | ((uint32_t)bi->hclen << 10);
                l = 14;
              } else if (j < 2 + bi->hclen + 4u) {
                const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                v = bi->cl_len[order[j - 2]];
                l = 3;
              } else {
                const uint32_t r = j - 2 - (bi->hclen + 4u);
                const uint32_t c = bi->rle_sym[r];
                const uint32_t cl = bi->cl_len[c];
                v = bi->cl_code[c];
                l = cl;
                if (c >= 16) {
                  v |= (uint64_t)bi->rle_extra[r] << cl;
                  l += rle_extra_bitlen((int)c);
                }
              }
            }
          }
        }
        val[k] = v;
        nb[k] = l;
      }
      pack_round(P, val, nb, false);
    }
  }

  if (type == 0) {
    // flush the header image, then raw bytes at the byte-aligned position (LD:1749)
    for (int k = 0; k < kPackItems; k++) {
      val[k] = 0;
      nb[k] = 0;
    }
    pack_round(P, val, nb, true);
    const uint64_t byte0 = P.bitpos >> 3;
    const uint8_t *src = B.vbuf + (size_t)s * (B.nvp + kVPad) + B.dlen + (bs - 1);
    const uint32_t size = be + 1 - bs;
    for (uint32_t i = tid; i < size; i += kPackThreads) out[byte0 + i] = src[i];
    P.bitpos += (uint64_t)size * 8;
    __syncthreads();
    if (!last) return;
    // restart the image at the new position for the trailer
    P.first_word = P.bitpos >> 5;
    P.block_first_word = P.first_word;
  } else {
    const uint32_t *tok = B.bufA + (size_t)s * B.nvp + (bs - 1);
    const uint32_t ntok = bi->ntok;
    const uint32_t nsym = ntok + 1;  // + EOB
    for (uint32_t r0 = 0; r0 < nsym; r0 += kPackRound) {
      // (a thread's tokens of the round, all loads in flight together: loaded where each is used, inside its branch,
      // they were four waits one after the other, each also behind the round before's stores)
      uint32_t tv[kPackItems];
#pragma unroll
      for (int k = 0; k < kPackItems; k++) {
        const uint32_t i = r0 + tid * kPackItems + k;
        tv[k] = tok[i < ntok ? i : (ntok ? ntok - 1 : 0u)];  // (a block has a token slot even when it has no token)
      }
#pragma unroll
      for (int k = 0; k < kPackItems; k++) {
        const uint32_t i = r0 + tid * kPackItems + k;
        uint64_t v = 0;
        uint32_t l = 0;
        if (i < nsym) {
          const uint32_t t = i < ntok ? tv[k] : 256u;
          if (t & kTokMatch) {
            const Code lc = length_code(tok_len(t));
            const Code dc = dist_code(tok_dist(t));
            uint32_t lcode, llen, dcode, dlen;
            if (type == 2) {
              const uint32_t le = s_lc[lc.code], de = s_dc[dc.code];
              lcode = le & 0xFFFFu;
              llen = le >> 16;
              dcode = de & 0xFFFFu;
              dlen = de >> 16;
            } else {  // fixed codes LD:2826-2849
              llen = fixed_lit_bits(lc.code);
              const uint32_t c = lc.code <= 279 ? (lc.code - 256) : (0xC0 + lc.code - 280);
              lcode = __brev(c) >> (32 - llen);
              dlen = 5;
              dcode = __brev(dc.code) >> 27;
            }
            v = lcode;
            l = llen;
            v |= (uint64_t)lc.eval << l;
            l += lc.ebits;
            v |= (uint64_t)dcode << l;
            l += dlen;
            v |= (uint64_t)dc.eval << l;
            l += dc.ebits;
          } else {
            if (type == 2) {
              const uint32_t le = s_lc[t];
              v = le & 0xFFFFu;
              l = le >> 16;
            } else {
              l = fixed_lit_bits(t);
              const uint32_t c = t <= 143 ? (0x30 + t) : (t <= 255 ? (0x190 + t - 144) : (t - 256));
              v = __brev(c) >> (32 - l);
            }
          }
        }
        val[k] = v;
        nb[k] = l;
      }
      pack_round(P, val, nb, (r0 + kPackRound >= nsym) && !last);
    }
    if (!last) return;
  }
  // stream epilogue: pad to a byte with ones (LD:796-812); zlib: Adler-32 big-endian (LD:2004-2020)
  {
    for (int k = 0; k < kPackItems; k++) {
      val[k] = 0;
      nb[k] = 0;
    }
    if (tid == 0) {
      const uint32_t pad = (uint32_t)((8 - P.bitpos % 8) % 8);
      val[0] = (1u << pad) - 1u;
      nb[0] = pad;
      if (B.fmt == 1) {
        const uint32_t ad = B.adler[s];
        val[1] = __builtin_bswap32(ad);
        nb[1] = 32;
      }
    }
    pack_round(P, val, nb, true);
  }
}

}  // namespace ldh
