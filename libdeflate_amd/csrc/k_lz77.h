// k_lz77.h -- north-star part (1): trigram chains, longest-match search, parse.
//
// Replaces GetBlockLZ77Result (LD:1241-1484) + LoadStringToTable (LD:1184-1197) and the
// dictionary chains of CreateDictionary (LD:531-595).
//
//   k_seg_links    levels 4-9, streams beyond 16 KiB: V = dict ++ input in segments of 32768
//                  positions, each sorted by exact trigram (Q1: the "hash" is the three bytes
//                  themselves) entirely inside LDS; prevd[v] = distance to the previous position
//                  with the same trigram (0 = none within 32768), across segments by look-up in
//                  the previous segment's last-occurrence list.
//   k_sort_links   the same for short streams (one workgroup per stream, sort through memory);
//                  for levels 1-3 the sorted array itself and its inverse; V alone for level 0.
//   k_match_lazy   levels 4-9.  Every position is inserted (Q4) and depth is always
//                  max_chain (Q3), so the best match at a position does not depend on
//                  the parse: one lane per position, window bytes + links staged in LDS and
//                  kept there from one tile of a stream to the next.
//   k_parse_lazy   the lazy state machine of LD:1318-1476 (Q5) as a successor function: one
//                  workgroup per deflate block, the walk found by relaxation over chunks.
//   k_lz_greedy    levels 1-3: insertion depends on the parse (Q4), one wavefront per stream.
#pragma once
#include "ldh_common.h"

namespace ldh {

// Blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous range
// of work items so that tiles of one stream reuse that XCD's L2.  Bijective for any total.
__device__ inline uint32_t xcd_remap(uint32_t bid, uint32_t total) {
  uint32_t q = total / 8, r = total % 8, xcd = bid % 8, j = bid / 8;
  uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

__device__ inline uint32_t trigram(const uint8_t *V, uint32_t v) {
  return ((uint32_t)V[v] << 16) | ((uint32_t)V[v + 1] << 8) | (uint32_t)V[v + 2];
}

// LDS reads at any byte offset.  gfx950 accepts misaligned ds_read_u16 / _b32 / _b64 (the
// compiler emits them for align-1 loads), but they are slow: with one misaligned ds_read_u16
// per chain step k_match_lazy took 91 ms instead of 33.  So: aligned reads + alignbyte.
// (the two bytes are joined by v_perm_b32, which the compiler does not see through: joined by
// shift-and-or it may fuse the byte reads into exactly that misaligned ds_read_u16 -- it did
// after a refactoring that changed nothing else, 27.7 -> 67 ms)
__device__ inline uint32_t lds_load2(const uint32_t *w, uint32_t byteoff) {
  const uint8_t *b = (const uint8_t *)w + byteoff;
  return __builtin_amdgcn_perm((uint32_t)b[1], (uint32_t)b[0], 0x0c0c0400u);
}
__device__ inline uint32_t lds_load4(const uint32_t *w, uint32_t byteoff) {
  uint32_t i = byteoff >> 2;
  return __builtin_amdgcn_alignbyte(w[i + 1], w[i], byteoff & 3);
}
__device__ inline uint64_t lds_load8(const uint32_t *w, uint32_t byteoff) {
  const uint32_t i = byteoff >> 2, sh = byteoff & 3;
  const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2];
  return ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32) | __builtin_amdgcn_alignbyte(w1, w0, sh);
}

// ------------------------------------------------------------------ k_sort_links
constexpr int kSortThreads = 1024;
constexpr int kSortThreadsNarrow = 256;
#ifndef LDH_SORT_NARROW_MAX
#define LDH_SORT_NARROW_MAX 16384
#endif
constexpr uint32_t kSortNarrowMaxLds = LDH_SORT_NARROW_MAX;  // streams up to about this long take the narrow version
#ifndef LDH_SORT_ITEMS
#define LDH_SORT_ITEMS 16
#endif
#ifndef LDH_LINK_ITEMS
#define LDH_LINK_ITEMS 8
#endif

// exclusive scan of 256 values held by threads 0..255 (one value each); result in out[256]
__device__ inline void scan256_exclusive(uint32_t val, uint32_t tid, uint32_t *out, uint32_t *wave_tot) {
  const uint32_t lane = tid & 63, wave = tid >> 6;
  // (DPP row shifts and broadcasts, six dependent vector instructions: the __shfl_up form is six trips through the
  // LDS crossbar, each waited for)
  const uint32_t inc = wave_inclusive_sum(val);
  if (tid < 256 && lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  if (tid < 256) {
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
    out[tid] = base + inc - val;
  }
  __syncthreads();
}

// V_LDS: the bytes being sorted also live in LDS (up to kSortLdsMaxV of them), so the
// per-element key fetches of passes 1-2 and the trigram compares of the link pass are LDS
// byte reads instead of fully divergent global loads (which bound the first version).
constexpr uint32_t kSortLdsMaxV = 131072 + 64;
// (lazy levels with streams beyond the narrow version's go to k_seg_links below, whatever their length)

// THREADS: 1024, or 256 for batches of short streams (a 4 KiB stream is half a tile of the wide
// version, and four times as many workgroups fit a CU)
template <bool V_LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_sort_links(Batch B) {
  constexpr int kSortThreads = THREADS, kSortWaves = THREADS / kWave;
  // elements per thread per tile: 16 for the wide version (fewer barriers per element: 12.7 -> 12.4 ms),
  // 8 for the narrow one (registers decide how many of its workgroups share a CU: 3.0 -> 2.6 ms at 4 KiB)
  constexpr int kSortItems = THREADS >= 1024 ? LDH_SORT_ITEMS : LDH_SORT_ITEMS / 2;
  extern __shared__ uint8_t ldsV[];
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_base[256];
  __shared__ uint32_t s_bprev[256];
  __shared__ uint16_t s_cnt[kSortWaves][256];
  __shared__ uint16_t s_pre[kSortWaves][256];
  __shared__ uint32_t s_tot[4];

  const uint32_t s = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = stream_len(B, s), D = B.dlen, nv_all = D + n, nv = nv_all;
  uint8_t *V = B.vbuf + (size_t)s * (B.nvp + kVPad);
  const uint8_t *in = B.in + B.in_off[s];
  uint32_t *bufA = B.bufA + (size_t)s * B.nvp;
  uint32_t *bufB = B.bufB + (size_t)s * B.nvp;
  uint16_t *prevd = B.prevd + (size_t)s * B.nvp;
  // where the sorted array ends up; greedy levels also get its inverse (rank of a position)
  // (greedy levels: entries of the final array carry the first twelve bytes of their position too)
  uint4 *srt2 = B.want_rank ? (uint4 *)B.bufC + (size_t)s * B.nvp : nullptr;
  uint32_t *rank = bufB;
#define SRT(r) (B.want_rank ? srt2[(r)].x : bufA[(r)])
#define VBYTE(i) (V_LDS ? (uint32_t)ldsV[(i)] : (uint32_t)V[(i)])

  if (tid < 256) s_hist[tid] = 0;
  for (uint32_t i = tid; i < kSortWaves * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
  __syncthreads();

  // V = dict ++ input ++ zeros, written as dwords (V is 4-byte aligned)
  const uint32_t total = (nv + kVPad) & ~3u;  // nv + kVPad <= nvp + kVPad
  // (whole dwords of the input where it is 4-byte aligned relative to V: the usual case)
  const bool in_aligned = (((uintptr_t)in - D) & 3) == 0;
  for (uint32_t w = tid * 4; w < total; w += kSortThreads * 4) {
    uint32_t word = 0;
    const uint32_t g0 = w;
    if (in_aligned && g0 >= D && g0 + 4 <= nv_all) {
      word = *(const uint32_t *)(in + (g0 - D));
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t gi = g0 + k;
        uint32_t byte = 0;
        if (gi < nv_all) byte = gi < D ? B.dict[gi] : in[gi - D];
        word |= byte << (8 * k);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (w + k < nv) atomicAdd(&s_hist[(word >> (8 * k)) & 255u], 1u);
    *(uint32_t *)(V + w) = word;
    if (V_LDS) *(uint32_t *)(ldsV + w) = word;
  }
  __syncthreads();
  if (nv == 0 || !B.need_lz) return;  // (stored blocks and huffman_only need V only)
#ifdef LDH_SORT_STOP  // profiling builds only (-DLDH_SORT_STOP=n: leave after stage n; results are wrong)
  if (LDH_SORT_STOP == 1) return;
#endif

  for (int pass = 0; pass < 3; pass++) {
#ifdef LDH_SORT_STOP
    if (pass + 1 >= LDH_SORT_STOP) return;
#endif
    const uint32_t shift = 2 - pass;  // digit of v is V[v + shift]
    const uint32_t *src = pass == 1 ? bufA : bufB;
    uint32_t *dst = pass == 1 ? bufB : bufA;
    {  // bin bases: histogram of V[shift .. shift+nv-1], exclusive scan
      uint32_t cnt = 0;
      if (tid < 256) {
        // s_hist counts V[0 .. nv-1]: take out what lies before `shift`, add what lies beyond nv
        const uint32_t m = shift < nv ? shift : nv, beyond = nv > shift ? nv : shift;
        cnt = s_hist[tid];
        for (uint32_t j = 0; j < m; j++) cnt += (VBYTE(beyond + j) == tid) - (VBYTE(j) == tid);
      }
      scan256_exclusive(cnt, tid, s_base, s_tot);
    }

    // A tile is kSortItems*1024 consecutive elements; wave w owns 64*kSortItems consecutive
    // ones of them, so (wave, sub-round, lane) order is element order and ranks stay stable.
    constexpr uint32_t kTileElems = kSortThreads * kSortItems;
    const uint32_t ntiles = (nv + kTileElems - 1) / kTileElems;
    // (the next tile's elements are fetched before this tile's scattered stores are issued:
    // behind them in the memory pipeline the loads would wait for all of them to drain)
    uint32_t en[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
      const uint32_t idx = wave * (64 * kSortItems) + lane + j * 64;
      en[j] = idx < nv ? (pass == 0 ? idx : src[idx]) : 0xFFFFFFFFu;
    }
    for (uint32_t tile = 0; tile < ntiles; tile++) {
      const uint32_t base_idx = tile * kTileElems + wave * (64 * kSortItems) + lane;
      uint32_t e[kSortItems], d[kSortItems], lrank[kSortItems];
#pragma unroll
      for (int j = 0; j < kSortItems; j++) {
        e[j] = en[j];
        const uint32_t idx = base_idx + kTileElems + j * 64;
        en[j] = idx < nv ? (pass == 0 ? idx : src[idx]) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int j = 0; j < kSortItems; j++) d[j] = e[j] != 0xFFFFFFFFu ? VBYTE(e[j] + shift) : 0;
#pragma unroll
      for (int j = 0; j < kSortItems; j++) {
        const bool valid = e[j] != 0xFFFFFFFFu;
        // lanes of this wave holding the same digit (stable rank = lanes below me)
        // (invalid lanes vote with digit 0; they are outside the initial mask, so a valid lane
        // never counts them.  A bit costs v_bfe_i32 + v_cmp + two v_bitop3.)
        const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
        uint32_t xlo = 0, xhi = 0;  // lanes whose digit differs from mine in some bit
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
          const uint32_t m = (uint32_t)((int32_t)(d[j] << (31 - bit)) >> 31);  // all ones: my bit is set
          const unsigned long long bal = __builtin_amdgcn_ballot_w64((int32_t)m < 0);
          // x |= bal ^ m, one v_bitop3 per half (truth table of a | (b ^ c))
          xlo = __builtin_amdgcn_bitop3_b32(xlo, (uint32_t)bal, m, 0xF6);
          xhi = __builtin_amdgcn_bitop3_b32(xhi, (uint32_t)(bal >> 32), m, 0xF6);
        }
        const uint32_t mlo = (uint32_t)vm & ~xlo, mhi = (uint32_t)(vm >> 32) & ~xhi;
        const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0));
        const uint32_t old = s_cnt[wave][d[j]];  // same digit in this wave's earlier sub-rounds
        lrank[j] = old + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) s_cnt[wave][d[j]] = (uint16_t)(old + (uint32_t)__popcll(mask));
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      if (tid < 256) {  // per digit: prefix over the waves; leaves s_cnt zeroed
        const uint32_t off0 = s_base[tid];
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; w++) {
          const uint32_t t = s_cnt[w][tid];
          s_cnt[w][tid] = 0;
          s_pre[w][tid] = (uint16_t)run;
          run += t;
        }
        s_bprev[tid] = off0;
        s_base[tid] = off0 + run;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kSortItems; j++)
        if (e[j] != 0xFFFFFFFFu) {
          const uint32_t at = s_bprev[d[j]] + s_pre[wave][d[j]] + lrank[j];
          if (pass == 2 && B.want_rank) {
            uint32_t w4[3];
#pragma unroll
            for (uint32_t q = 0; q < 3; q++) {
              const uint32_t o = e[j] + 4 * q;  // (V is zero padded: kVPad bytes)
              w4[q] = V_LDS ? lds_load4((const uint32_t *)ldsV, o)
                            : VBYTE(o) | (VBYTE(o + 1) << 8) | (VBYTE(o + 2) << 16) | (VBYTE(o + 3) << 24);
            }
            srt2[at] = make_uint4(e[j], w4[0], w4[1], w4[2]);
            if (V_LDS) dst[at] = e[j];  // (the compact copy the un-sort below reads)
          } else {
            dst[at] = e[j];
          }
        }
      // next tile: its s_cnt updates alias neither s_pre nor s_bprev; the barrier after them
      // orders the reads above before the next rewrite.
    }
    __syncthreads();
  }

#ifdef LDH_SORT_STOP
  if (LDH_SORT_STOP == 4) return;
#endif
  if (B.want_rank) {
    // greedy levels walk the sorted array itself (k_lz_greedy): no links, only the inverse permutation
    if (V_LDS) {
      // like the links below: scattering 4-byte ranks by position straight to memory took longer than
      // the three sort passes together; V's LDS (free now) takes the ranks of a quarter of the positions
      // at a time and they are copied out in order
      uint32_t *l32 = (uint32_t *)ldsV;
      const uint32_t H4 = (B.sort_lds / 4) & ~3u;
      constexpr uint32_t kItems = LDH_LINK_ITEMS;
      for (uint32_t h0 = 0; h0 < nv; h0 += H4) {
        for (uint32_t r0 = tid; r0 < nv; r0 += kSortThreads * kItems) {
          uint32_t v[kItems];
#pragma unroll
          for (uint32_t j = 0; j < kItems; j++) {
            const uint32_t r = r0 + j * kSortThreads;
            v[j] = r < nv ? bufA[r] : 0xFFFFFFFFu;
          }
#pragma unroll
          for (uint32_t j = 0; j < kItems; j++)
            if (v[j] != 0xFFFFFFFFu && v[j] - h0 < H4) l32[v[j] - h0] = r0 + j * kSortThreads;
        }
        __syncthreads();
        const uint32_t cnt = nv - h0 < H4 ? nv - h0 : H4;
        for (uint32_t i = tid; i < cnt; i += kSortThreads) rank[h0 + i] = l32[i];
        __syncthreads();
      }
    } else {
      for (uint32_t r = tid; r < nv; r += kSortThreads) rank[SRT(r)] = r;
    }
    return;
  }
  // links: neighbours in sorted order that share the trigram
  if (V_LDS) {
    // Scattering 2-byte links by position straight to memory cost as much as the whole sort.
    // Instead: (a) link values in sorted order, coalesced, into the free ping-pong buffer;
    // (b) V is no longer needed, so its LDS holds the links of 65536 positions at a time:
    // scatter there, then copy out in order.
    // (four elements per thread and trip, loads first: one workgroup per CU has nothing else to
    // hide their latency with)
    uint16_t *dl = (uint16_t *)bufB;
    constexpr uint32_t kLinkItems = LDH_LINK_ITEMS;
    for (uint32_t r0 = tid; r0 < nv; r0 += kSortThreads * kLinkItems) {
      uint32_t v[kLinkItems], u[kLinkItems];
#pragma unroll
      for (uint32_t j = 0; j < kLinkItems; j++) {
        const uint32_t r = r0 + j * kSortThreads;
        v[j] = r < nv ? bufA[r] : 0u;
        u[j] = (r > 0 && r < nv) ? bufA[r - 1] : 0u;
      }
#pragma unroll
      for (uint32_t j = 0; j < kLinkItems; j++) {
        const uint32_t r = r0 + j * kSortThreads;
        uint32_t dlt = 0;
        if (r > 0 && r < nv) {
          const uint32_t tu = lds_load4((const uint32_t *)ldsV, u[j]) << 8;  // three bytes (V is zero padded)
          const uint32_t tv = lds_load4((const uint32_t *)ldsV, v[j]) << 8;
          if (tu == tv && v[j] - u[j] <= kWindow) dlt = v[j] - u[j];
        }
        if (r < nv) dl[r] = (uint16_t)dlt;
      }
    }
    __syncthreads();
    uint16_t *l16 = (uint16_t *)ldsV;
    const uint32_t H = (B.sort_lds / 2) & ~1u;  // positions whose links fit V's LDS (>= 24, even)
    for (uint32_t h0 = 0; h0 < nv; h0 += H) {
      for (uint32_t r0 = tid; r0 < nv; r0 += kSortThreads * kLinkItems) {
        uint32_t v[kLinkItems], l[kLinkItems];
#pragma unroll
        for (uint32_t j = 0; j < kLinkItems; j++) {
          const uint32_t r = r0 + j * kSortThreads;
          v[j] = r < nv ? bufA[r] : 0xFFFFFFFFu;
          l[j] = r < nv ? dl[r] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kLinkItems; j++)
          if (v[j] != 0xFFFFFFFFu && v[j] - h0 < H) l16[v[j] - h0] = (uint16_t)l[j];
      }
      __syncthreads();
      const uint32_t cnt = nv - h0 < H ? nv - h0 : H;  // prevd rows and H are even: 4-byte aligned
      const uint32_t *src32 = (const uint32_t *)l16;
      uint32_t *dst32 = (uint32_t *)(prevd + h0);
      for (uint32_t i = tid; i < (cnt + 1) / 2; i += kSortThreads) dst32[i] = src32[i];
      __syncthreads();
    }
  } else {
    for (uint32_t r = tid; r < nv; r += kSortThreads) {
      uint32_t v = SRT(r);
      uint32_t dlt = 0;
      if (r > 0) {
        uint32_t u = SRT(r - 1);
        const uint32_t tu = (VBYTE(u) << 16) | (VBYTE(u + 1) << 8) | VBYTE(u + 2);
        const uint32_t tv = (VBYTE(v) << 16) | (VBYTE(v + 1) << 8) | VBYTE(v + 2);
        if (tu == tv) {
          uint32_t d = v - u;
          if (d <= kWindow) dlt = d;
        }
      }
      prevd[v] = (uint16_t)dlt;
    }
  }
#undef VBYTE
#undef SRT
}

// ------------------------------------------------------------------ k_seg_links
// The links of the lazy levels for anything longer than the narrow sort's streams, with the sort held
// entirely in LDS.  V is cut into segments of 32768 positions.  A segment's positions (16-bit, relative
// to the segment) are sorted by trigram where they lie: every thread holds its 32 elements in registers
// across the workgroup's rank computation, so a pass scatters in place and nothing of the sort touches
// memory.  Neighbours in sorted order give the links inside the segment; they are scattered by position
// into the same LDS and leave it coalesced.  What is missing then is the link of a trigram's first
// occurrence in a segment: chains reach back 32768 positions (LD:1365), i.e. into the segment before and
// no further, and there only the trigram's last occurrence matters.  So every segment leaves the list of
// its last occurrences {trigram, position}, sorted by trigram as the sort left them, and the next
// segment's first occurrences are looked up in it (binary search, list staged through LDS a piece at a
// time, the searches of a wave gathered into full rows of 64 first).  A workgroup takes a run of
// consecutive segments of one stream so that the list goes from one iteration to the next through memory
// it alone touches (the segment's slice of bufA / bufB, unused until k_match_lazy); a run that does not
// begin the stream sorts the segment before it once more just for that list.  The host picks the run
// length: whole streams when the batch has workgroups enough without cutting them (no segment sorted
// twice), else kSegRun segments (any length in parallel at 1.25x the work).
constexpr uint32_t kSegG = 32768;
constexpr uint32_t kSegRun = 4;
constexpr int kSegThreads = 1024;
constexpr int kSegWaves = kSegThreads / kWave;
constexpr int kSegItems = kSegG / kSegThreads;
constexpr uint32_t kSegPiece = 4096;
constexpr uint32_t kSegQueue = 128;
constexpr uint32_t kSegLdsV = kSegG + 64;                // the segment's bytes + kVPad, 16-byte granular
constexpr uint32_t kSegLdsBytes = kSegLdsV + 2 * kSegG;  // dynamic LDS: bytes, then elements / links

// trigram at byte offset v of the staged bytes, first byte most significant (the order of the sort)
__device__ inline uint32_t seg_key(const uint32_t *w, uint32_t v) {
  return __builtin_amdgcn_perm(0u, lds_load4(w, v), 0x0c000102u);
}

__global__ __launch_bounds__(kSegThreads) void k_seg_links(Batch B, uint32_t runs, uint32_t run_len) {
  extern __shared__ uint8_t seg_lds[];
  uint8_t *Vs = seg_lds;
  const uint32_t *Vw = (const uint32_t *)seg_lds;
  uint16_t *E = (uint16_t *)(seg_lds + kSegLdsV);
  __shared__ uint16_t s_cnt[kSegWaves][256];
  __shared__ __attribute__((aligned(16))) uint16_t s_pre[kSegWaves][256];
  __shared__ uint32_t s_tot[16];
  __shared__ uint32_t s_top;
  __shared__ uint32_t s_lcnt[kSegItems * kSegWaves];  // list entries per (row of 1024, wave), then where they go
  __shared__ uint32_t sK[kSegPiece];  // a piece of the previous segment's list
  __shared__ uint16_t sP[kSegPiece];
  __shared__ uint16_t qv[kSegWaves][kSegQueue];  // look-ups (positions) waiting for a full row, per wave

  const uint32_t s = blockIdx.x / runs, run = blockIdx.x % runs;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = stream_len(B, s), D = B.dlen, nv_all = D + n;
  uint8_t *Vg = B.vbuf + (size_t)s * (B.nvp + kVPad);
  const uint8_t *in = B.in + B.in_off[s];
  if (nv_all == 0) {  // only the zeros after the end
    if (run == 0 && tid < kVPad / 4) ((uint32_t *)Vg)[tid] = 0;
    return;
  }
  const uint32_t nseg = (nv_all + kSegG - 1) / kSegG;
  const uint32_t g_first = run * run_len;
  if (g_first >= nseg) return;
  const uint32_t g_end = g_first + run_len < nseg ? g_first + run_len : nseg;
  const bool in_aligned = (((uintptr_t)in - D) & 3) == 0;
  for (uint32_t i = tid; i < kSegWaves * 256; i += kSegThreads) (&s_cnt[0][0])[i] = 0;
  if (tid == 0) s_top = 0;
  __syncthreads();

  uint32_t nprev = 0;  // entries of the list the previous iteration left
  for (uint32_t g = g_first ? g_first - 1 : 0; g < g_end; g++) {
    const bool warm = g < g_first;        // sorted for its list only
    const bool want_list = g + 1 < g_end; // the next iteration reads this segment's list
    const uint32_t lo = g * kSegG, m = nv_all - lo < kSegG ? nv_all - lo : kSegG;
    uint32_t *listK = B.bufA + (size_t)s * B.nvp + lo;             // at most m entries of 4 bytes
    uint16_t *listP = (uint16_t *)(B.bufB + (size_t)s * B.nvp + lo);  // and of 2
    const uint32_t *prevK = listK - (g ? kSegG : 0u);
    const uint16_t *prevP = (const uint16_t *)(B.bufB + (size_t)s * B.nvp + lo - (g ? kSegG : 0u));

    // bytes [lo, lo + m + kVPad) of V = dict ++ input ++ zeros; to memory what this segment owns (the
    // bytes after a segment are the next one's, the zeros after the end the last one's)
    {
      const uint32_t total = (m + kVPad) & ~3u;
      const uint32_t wlimit = warm ? 0u : (lo + m == nv_all ? total : m);
      // whole dwords of the dictionary and of the input (where it is 4-byte aligned relative to V, the
      // usual case): all of a thread's loads in flight at once -- one workgroup per CU has nothing else
      // to hide them behind (a load, a store and the next load in turn took 0.9 of the kernel's 11.6 ms)
      const uint32_t nw = total / 4;
      uint32_t any_bits = 0;
      const bool dict_al = (((uintptr_t)B.dict) & 3) == 0;
      // 1: a dword of the dictionary, 2: of the input, 0: bytes (an edge, or nothing aligned)
      auto kind = [&](uint32_t idx) -> uint32_t {
        const uint32_t g0 = lo + idx * 4;
        if (dict_al && g0 + 4 <= D) return 1u;
        if (in_aligned && g0 >= D && g0 + 4 <= nv_all) return 2u;
        return 0u;
      };
      uint32_t ts = tid;  // (opaque: or the addresses below are kept, spilled, for the whole kernel)
      asm volatile("" : "+v"(ts));
      {
        constexpr int kW = kSegG / 4 / kSegThreads;  // 8
        uint32_t wd[kW + 1];
#pragma unroll
        for (int i = 0; i <= kW; i++) {
          const uint32_t idx = i * kSegThreads + ts, g0 = lo + idx * 4;
          const uint32_t kd = idx < nw ? kind(idx) : 0u;
          wd[i] = 0;
          if (kd == 1) wd[i] = *(const uint32_t *)(B.dict + g0);
          if (kd == 2) wd[i] = *(const uint32_t *)(in + (g0 - D));
        }
#pragma unroll
        for (int i = 0; i <= kW; i++) {
          const uint32_t idx = i * kSegThreads + ts;
          if (idx < nw && kind(idx)) {
            if (idx * 4 < wlimit) *(uint32_t *)(Vg + lo + idx * 4) = wd[i];
            *(uint32_t *)(Vs + idx * 4) = wd[i];
          }
          any_bits |= wd[i];
        }
      }
      for (uint32_t idx = ts; idx < nw; idx += kSegThreads) {  // the rest byte by byte
        if (kind(idx)) continue;
        uint32_t word = 0;
        const uint32_t w = idx * 4, g0 = lo + w;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t gi = g0 + k;
          uint32_t byte = 0;
          if (gi < nv_all) byte = gi < D ? B.dict[gi] : in[gi - D];
          word |= byte << (8 * k);
        }
        if (w < wlimit) *(uint32_t *)(Vg + lo + w) = word;
        *(uint32_t *)(Vs + w) = word;
        any_bits |= word;
      }
      // (text: no byte of the segment has its top bit set, and the passes rank by seven bits)
      if (__builtin_amdgcn_ballot_w64((any_bits & 0x80808080u) != 0) && lane == 0) s_top = 1;
    }
    __syncthreads();
    const bool top_bit = s_top != 0;

#ifdef LDH_SEG_STOP  // profiling builds only (-DLDH_SEG_STOP=n: skip what follows stage n; results are wrong)
    if (LDH_SEG_STOP == 1) continue;
#endif
    // three passes of a stable counting sort by V[v+2], V[v+1], V[v]; wave w owns the elements
    // [w*per*64, (w+1)*per*64), row j of them is one element per lane.  FULL: a whole segment, nothing to
    // check per element or per row (a sixth of the instructions, and the rows' digit reads are no longer
    // fenced in by branches).
    auto sort_passes = [&](auto full_tag, auto bits_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      constexpr int BITS = decltype(bits_tag)::value;
      const uint32_t per = FULL ? (uint32_t)kSegItems : (m + kSegThreads - 1) / kSegThreads;
#pragma unroll 1
      for (int pass = 0; pass < 3; pass++) {
#ifdef LDH_SEG_STOP
        if (pass + 2 > LDH_SEG_STOP) break;
#endif
        const uint32_t shift = 2 - pass;
        // (opaque: what is derived from the thread index below is invariant, and hoisted out of both loops
        // the 32 element indices alone keep 32 registers for the whole kernel)
        uint32_t t_ = tid;
        asm volatile("" : "+v"(t_));
        const uint32_t lane = t_ & 63, wave = t_ >> 6;
        uint32_t e[kSegItems];  // position (0xFFFF: none) | its digit << 16, then | its rank among the wave's same digits << 16
#pragma unroll
        for (int j = 0; j < kSegItems; j++) {
          const uint32_t idx = (wave * per + j) * 64 + lane;
          e[j] = (FULL || ((uint32_t)j < per && idx < m)) ? (pass == 0 ? idx : (uint32_t)E[idx]) : 0xFFFFu;
        }
#pragma unroll
        for (int j = 0; j < kSegItems; j++)
          if (FULL || e[j] != 0xFFFFu) e[j] |= (uint32_t)Vs[e[j] + shift] << 16;
        // (one row after the other: two rows' ballots interleaved, to fill the wait states between a
        // compare and the use of its mask, measured slower: 11.6 -> 12.0 ms)
#pragma unroll
        for (int j = 0; j < kSegItems; j++) {
          if (FULL || (uint32_t)j < per) {
            const bool valid = FULL || (e[j] & 0xFFFFu) != 0xFFFFu;
            const uint32_t d = e[j] >> 16;
            uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;  // lanes whose digit agrees with mine in every bit so far
#pragma unroll
            for (int bit = 0; bit < BITS; bit++) {
              const uint32_t mk = (uint32_t)((int32_t)(d << (31 - bit)) >> 31);  // all ones: my bit is set
              const unsigned long long bal = __builtin_amdgcn_ballot_w64((int32_t)mk < 0);
              // m &= ~(bal ^ mk), one v_bitop3 per half (truth table of a & ~(b ^ c))
              mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bal, mk, 0x90);
              mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bal >> 32), mk, 0x90);
            }
            if (!FULL) {  // (lanes without an element vote with digit 0 and are masked out here)
              const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
              mlo &= (uint32_t)vm;
              mhi &= (uint32_t)(vm >> 32);
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0));
            const uint32_t old = s_cnt[wave][d];  // same digit in this wave's earlier rows
            e[j] = (e[j] & 0xFFFFu) | ((old + rank) << 16);  // (at most 2047)
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) s_cnt[wave][d] = (uint16_t)(old + (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi));
            __builtin_amdgcn_wave_barrier();
          }
        }
        __syncthreads();  // (also: every element of the pass has been read, the scatter may overwrite them)
        // per digit: where each wave's elements go = elements with smaller digits + those of earlier
        // waves (one table for the scatter to read, not a base and a prefix); leaves s_cnt zeroed
        {
          uint32_t tot = 0;
          if (tid < 256) {
#pragma unroll
            for (int w = 0; w < kSegWaves; w++) tot += s_cnt[w][tid];
          }
          uint32_t inc = tot;
          if (tid < 256) {
            inc = wave_inclusive_sum(inc);
            if (lane == 63) s_tot[wave] = inc;
          }
          __syncthreads();
          if (tid < 256) {
            uint32_t at = inc - tot;
            for (uint32_t w = 0; w < wave; w++) at += s_tot[w];
#pragma unroll
            for (int w = 0; w < kSegWaves; w++) {  // (the counts are read a second time: 16 registers less)
              const uint32_t c = s_cnt[w][tid];
              s_cnt[w][tid] = 0;
              s_pre[w][tid] = (uint16_t)at;  // (at most 32767)
              at += c;
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < kSegItems; j++)
          if (FULL || (e[j] & 0xFFFFu) != 0xFFFFu) {
            // (the digit is read again: kept in registers, four to one, the packing and unpacking cost
            // more vector instructions than the LDS read saves -- 11.1 -> 11.6 ms)
            const uint32_t d = Vs[(e[j] & 0xFFFFu) + shift];
            E[(uint32_t)s_pre[wave][d] + (e[j] >> 16)] = (uint16_t)e[j];
          }
        __syncthreads();
      }
    };
    if (m == kSegG && !top_bit)
      sort_passes(std::true_type{}, std::integral_constant<int, 7>{});
    else if (m == kSegG)
      sort_passes(std::true_type{}, std::integral_constant<int, 8>{});
    else
      sort_passes(std::false_type{}, std::integral_constant<int, 8>{});

    // neighbours in sorted order: element r = j*1024 + tid; res = position | link << 15 | first << 30 | last << 31.
    // A lane gets its neighbours' trigrams from the lanes beside it; what the ends of a row of 64 need from
    // the rows beside it is put into LDS beforehand (the queues' space, idle until the look-ups below), so
    // that no lane branches here.
#ifdef LDH_SEG_STOP
    if (LDH_SEG_STOP <= 4) continue;
#endif
    // (in s_pre's space, idle between the passes and the next segment's: the look-ups below read the rows' trigram
    // ranges again, when the queues are in use)
    static_assert(sizeof(s_pre) >= kSegItems * kSegWaves * (4 + 4 + 2), "row tables in s_pre");
    uint32_t *rowK0 = (uint32_t *)&s_pre[0][0], *rowK1 = rowK0 + kSegItems * kSegWaves;  // trigram of a row's first / last element
    uint16_t *rowP1 = (uint16_t *)(rowK1 + kSegItems * kSegWaves);                       // position of its last element
    if (tid < kSegItems * kSegWaves) {
      const uint32_t r0 = tid * 64, r1 = r0 + 63;
      rowK0[tid] = r0 < m ? seg_key(Vw, (uint32_t)E[r0]) : 0xFFFFFFFEu;
      const uint32_t p1 = r1 < m ? (uint32_t)E[r1] : 0u;
      rowK1[tid] = r1 < m ? seg_key(Vw, p1) : 0xFFFFFFFEu;
      rowP1[tid] = (uint16_t)p1;
    }
    __syncthreads();
    // (the trigram of a position is read again where it is needed below: 32 registers less)
    uint32_t res[kSegItems];
    uint32_t t_ = tid;  // (opaque, as in the passes)
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < kSegItems; j++) {
      const uint32_t r = j * kSegThreads + t_, row = j * kSegWaves + (t_ >> 6);
      const bool valid = r < m;
      const uint32_t v = valid ? (uint32_t)E[r] : 0u;
      const uint32_t k = valid ? seg_key(Vw, v) : 0xFFFFFFFFu;
      // (the lanes beside this one by DPP wave shifts, not through the LDS crossbar; lanes 0 / 63 take the rows' values)
      const uint32_t pv_in = wave_shr1(v), pk_in = wave_shr1(k), nk_in = wave_shl1(k);
      const uint32_t pv_row = row ? (uint32_t)rowP1[row - 1] : 0u, pk_row = row ? rowK1[row - 1] : 0xFFFFFFFEu;
      const uint32_t nk_row = row + 1 < kSegItems * kSegWaves ? rowK0[row + 1] : 0xFFFFFFFEu;
      const uint32_t pv = lane == 0 ? pv_row : pv_in, pk = lane == 0 ? pk_row : pk_in, nk = lane == 63 ? nk_row : nk_in;
      const bool same = valid && pk == k;
      const uint32_t link = same ? v - pv : 0u;
      const uint32_t first = (valid && !same) ? 1u : 0u, last = (valid && nk != k) ? 1u : 0u;
      res[j] = v | (link << 15) | (first << 30) | (last << 31);
      // (opaque on purpose: the optimiser sees through the packing and keeps the four fields of all 32
      // elements in registers of their own -- 200 spilled registers)
      asm volatile("" : "+v"(res[j]));
    }
    __syncthreads();  // the elements have been read: their LDS now takes the links by position

    if (!warm) {
      asm volatile("" : "+v"(t_));
#pragma unroll
      for (int j = 0; j < kSegItems; j++)
        if ((uint32_t)j * kSegThreads + t_ < m) E[res[j] & 0x7FFFu] = (uint16_t)((res[j] >> 15) & 0x7FFFu);
    }
    // this segment's list, in sorted order
    if (want_list) {
      asm volatile("" : "+v"(t_));
#pragma unroll
      for (int j = 0; j < kSegItems; j++) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64((res[j] >> 31) != 0);
        if ((t_ & 63) == 0) s_lcnt[j * kSegWaves + (t_ >> 6)] = (uint32_t)__popcll(bal);
      }
    }
    __syncthreads();
    uint32_t ntot = 0;
    {
      uint32_t val = 0, inc = 0;
      if (want_list && tid < kSegItems * kSegWaves) {
        val = s_lcnt[tid];
        inc = val;
        inc = wave_inclusive_sum(inc);
        if (lane == 63) s_tot[wave] = inc;
      }
      __syncthreads();
      if (want_list) {
        if (tid < kSegItems * kSegWaves) {
          uint32_t base = 0;
          for (uint32_t w = 0; w < wave; w++) base += s_tot[w];
          s_lcnt[tid] = base + inc - val;
        }
        for (uint32_t w = 0; w < kSegItems * kSegWaves / 64; w++) ntot += s_tot[w];
      }
      __syncthreads();
    }
    if (want_list) {
      asm volatile("" : "+v"(t_));
#pragma unroll
      for (int j = 0; j < kSegItems; j++) {
        const bool l = (res[j] >> 31) != 0;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(l);
        if (l) {
          const uint32_t slot = s_lcnt[j * kSegWaves + (t_ >> 6)] +
                                __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
          listK[slot] = seg_key(Vw, res[j] & 0x7FFFu);
          listP[slot] = (uint16_t)(res[j] & 0x7FFFu);
        }
      }
    }

    // first occurrences: look for the trigram in the list of the segment before
#ifdef LDH_SEG_STOP
    if (LDH_SEG_STOP <= 5) nprev = 0;
#endif
    if (!warm && nprev) {
#pragma unroll 1
      for (uint32_t p0 = 0; p0 < nprev; p0 += kSegPiece) {
        const uint32_t pc = nprev - p0 < kSegPiece ? nprev - p0 : kSegPiece;
        {  // (all of a thread's loads in flight together: in a loop, each trip waited for its own)
          constexpr int kPer = kSegPiece / kSegThreads;
          uint32_t kk[kPer], pp[kPer];
          uint32_t tl = t_;
          asm volatile("" : "+v"(tl));
#pragma unroll
          for (int r = 0; r < kPer; r++) {
            const uint32_t i = tl + r * kSegThreads, ic = i < pc ? i : pc - 1;
            kk[r] = prevK[p0 + ic];
            pp[r] = prevP[p0 + ic];
          }
#pragma unroll
          for (int r = 0; r < kPer; r++) {
            const uint32_t i = tl + r * kSegThreads;
            if (i < pc) {
              sK[i] = kk[r];
              sP[i] = (uint16_t)pp[r];
            }
          }
        }
        __syncthreads();
        const uint32_t kmin = sK[0], kmax = sK[pc - 1];
        // one search per lane: the first occurrences of a wave are gathered in its queue until 64 wait
        // (The queue holds positions only; the trigram is read here, by 64 lanes at once, not in the gathering loop,
        // where it was a dependent LDS round trip per row.  The search narrows by quarters: three independent reads a
        // step, seven steps for a piece of 4096 instead of twelve dependent ones.)
        auto search = [&](uint32_t wq, uint32_t at, bool act) {
#if defined(LDH_SEG_STOP) && LDH_SEG_STOP == 6  // profiling builds only: everything but the searches themselves
          return;
#endif
          const uint32_t v = act ? (uint32_t)qv[wq][at] : 0u;
          const uint32_t k = seg_key(Vw, v);
          uint32_t b = 0, len = pc;
          while (len > 3) {  // (uniform: pc is)
            const uint32_t q = len >> 2;
            const uint32_t a1 = sK[b + q], a2 = sK[b + 2 * q], a3 = sK[b + 3 * q];
            uint32_t adv = a1 <= k ? q : 0u;
            adv = a2 <= k ? 2 * q : adv;
            adv = a3 <= k ? 3 * q : adv;
            b += adv;
            len -= 3 * q;  // (the last quarter's length: at least as long as the others, and still inside the piece)
          }
          while (len > 1) {
            const uint32_t half = len >> 1;
            if (sK[b + half] <= k) b += half;
            len -= half;
          }
          if (act && sK[b] == k) {
            const uint32_t u = sP[b];
            if (u >= v) E[v] = (uint16_t)(kSegG + v - u);  // distance <= 32768
          }
        };
        // (the elements are in sorted order: a row whose trigrams lie outside the piece's has nothing to look up in it.
        // Lane j tests row j of this wave; the rows to look at are then a wave-uniform mask.)
        uint32_t rows_in;
        {
          uint32_t tq = t_;
          asm volatile("" : "+v"(tq));
          const uint32_t jr = tq & 31, wq = tq >> 6;
          const bool in = !(rowK1[jr * kSegWaves + wq] < kmin || rowK0[jr * kSegWaves + wq] > kmax);
          rows_in = (uint32_t)__builtin_amdgcn_ballot_w64(in);
        }
        uint32_t cnt = 0;
#if defined(LDH_SEG_STOP) && LDH_SEG_STOP == 7  // profiling builds only: the pieces' staging and barriers alone
        rows_in = 0;
#endif
#pragma unroll
        for (int j = 0; j < kSegItems; j++) {
          // (thread index opaque per element: the queue addresses derived from it are two instructions
          // to make; hoisted, they were spilled and reloaded from scratch memory 160 times per piece)
          uint32_t tq = t_;
          asm volatile("" : "+v"(tq));
          const uint32_t lq = tq & 63, wq = tq >> 6;
          if (!((rows_in >> j) & 1u)) continue;
          // (every first occurrence of a row that touches the piece's range: the few whose trigram lies outside it are
          // looked up in vain, which costs less than reading every trigram here to tell)
          const bool f = ((res[j] >> 30) & 1u) != 0;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
          if (bal) {
            if (f) {
              const uint32_t slot = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
              qv[wq][slot] = (uint16_t)(res[j] & 0x7FFFu);
            }
            cnt += (uint32_t)__popcll(bal);
            __builtin_amdgcn_wave_barrier();
            if (cnt >= 64) {
              search(wq, cnt - 64 + lq, true);
              cnt -= 64;
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
        if (cnt) search(wave, lane, lane < cnt);
        __syncthreads();
      }
    }
    __syncthreads();
    if (!warm) {  // links out, in order (rows of prevd and lo are 4-byte aligned)
      const uint32_t *src32 = (const uint32_t *)E;
      uint32_t *dst32 = (uint32_t *)(B.prevd + (size_t)s * B.nvp + lo);
      asm volatile("" : "+v"(t_));
      for (uint32_t i = t_; i < (m + 1) / 2; i += kSegThreads) dst32[i] = src32[i];
    }
    if (tid == 0) s_top = 0;  // (read before the passes; set again behind the next staging's barrier)
    __syncthreads();
    nprev = want_list ? ntot : 0u;
  }
}

// ------------------------------------------------------------------ k_match_lazy
#ifndef LDH_MATCH_TILE
#define LDH_MATCH_TILE 16384
#endif
#ifndef LDH_MATCH_THREADS
#define LDH_MATCH_THREADS 1024
#endif
#ifndef LDH_MATCH_MARGIN
#define LDH_MATCH_MARGIN 4096
#endif
// kTile + kMatchMargin: a multiple of 4096, at most 20480 (LDS); a tile holds one deflate-block end at most.
// kMatchMargin: positions kept in LDS below the window of a tile's first position, so that the walks of the last
// kMatchMargin positions of the tile before it can go on across the change of tiles instead of being waited for.
constexpr uint32_t kTile = LDH_MATCH_TILE;
constexpr uint32_t kMatchMargin = LDH_MATCH_MARGIN;
constexpr int kMatchThreads = LDH_MATCH_THREADS;
constexpr int kMatchThreadsNarrow = 256;
constexpr uint32_t kMatchNarrowMaxV = 16384;  // dictionary + stream up to this long: the narrow version, LDS sized to the batch
constexpr uint32_t kLdsByteCap = kWindow + kMatchMargin + kTile + 288;  // window + margin + tile + max match + slack
constexpr uint32_t kLdsPrevCap = kWindow + kMatchMargin + kTile + 8;
static_assert(kLdsByteCap + 2 * kLdsPrevCap <= 160 * 1024 - 64, "k_match_lazy's LDS");
constexpr uint32_t kMatchLdsBytes = kLdsByteCap + 2 * kLdsPrevCap;
#ifndef LDH_MATCH_STEPS
#define LDH_MATCH_STEPS 8
#endif

#ifndef LDH_MATCH_WALKS
#define LDH_MATCH_WALKS 1
#endif
#ifndef LDH_REFILL_MIN_DEEP
#define LDH_REFILL_MIN_DEEP 12
#endif
constexpr int kRefillMinDeep = LDH_REFILL_MIN_DEEP;  // REFILL_MIN of k_match_lazy for levels 7-9
#ifndef LDH_REFILL_MIN_SHALLOW
#define LDH_REFILL_MIN_SHALLOW 1
#endif
constexpr int kRefillMinShallow = LDH_REFILL_MIN_SHALLOW;  // ... and for levels 4-6
constexpr uint32_t kMatchSteps = LDH_MATCH_STEPS;  // chain steps between two looks at the long paths

// One lane works on one position at a time and walks its chain one candidate per loop
// trip; lanes whose chain has ended take the next position of the tile from a workgroup
// counter (chains are 0..max_chain long, so a static position->lane map idles most lanes).
// REFILL_MIN: idle lanes a wave waits for before it fetches new positions.  1 (levels 4-6): every
// round, with no branch around the refill.  At max_chain 256 and more a round ends only about four
// walks, and running the refill for them every round costs more than letting them wait.
// THREADS: 1024, or 256 for batches of short streams (a 4 KiB stream gives 1024 lanes four positions each, most of
// them spent in the drain at the tile's end; with its few KB of LDS eight of the small workgroups share a CU).
// lds_byte_cap: bytes of LDS set aside for the staged bytes (a multiple of 4); the links follow them.
template <bool HAS_DICT, int REFILL_MIN, int THREADS>
__global__ __launch_bounds__(THREADS) void k_match_lazy(Batch B, uint32_t tiles_per_stream, uint32_t lds_byte_cap,
                                                        uint32_t tiles_per_wg) {
  constexpr int kMatchThreads = THREADS;
  extern __shared__ uint32_t smem[];
  uint32_t *lbytes = smem;                                  // lds_byte_cap bytes
  uint16_t *lprev = (uint16_t *)(smem + lds_byte_cap / 4);  // window + tile + 8 entries
  __shared__ uint32_t s_next;

  // A workgroup takes `tiles_per_wg` consecutive tiles of one stream: from the second one on, the window it needs is
  // mostly in LDS already (the last 32768 positions of what the tile before had staged), is moved down there, and
  // only the new tile's bytes and links come from memory (60 KB instead of 160: staging was 1.6 of the kernel's
  // 21.6 ms per GiB).
  const uint32_t runs_per_stream = (tiles_per_stream + tiles_per_wg - 1) / tiles_per_wg;
  const uint32_t work = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t s = work / runs_per_stream, tile_first = (work % runs_per_stream) * tiles_per_wg;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t n = stream_len(B, s), D = HAS_DICT ? B.dlen : 0;
  if (tile_first * kTile >= n) return;
  const uint8_t *V = B.vbuf + (size_t)s * (B.nvp + kVPad);
  const uint16_t *prevd = B.prevd + (size_t)s * B.nvp;
  uint32_t *mout = B.bufB + (size_t)s * B.nvp;
  const LevelCfg cfg = level_cfg(B.level);
  const uint32_t nice = (uint32_t)cfg.nice;
  const uint32_t alloc_end = D + n + kVPad;  // bytes valid in vbuf

  // the current tile: stream positions [q0, q1) 0-based = V positions [v0, v1); LDS holds bytes
  // [wbase, v1 + 280) (zero beyond the allocated pad) and links [wbase, v1)
  uint32_t q0 = tile_first * kTile, q1 = 0, v0 = 0, v1 = 0, wbase = 0, be_lo = 0, be_hi = 0;
  // dwords [b_from, ..) of the bytes and [p_from, ..) of the links, relative to wbase, from memory
  auto stage = [&](uint32_t b_from, uint32_t p_from) {
    const uint32_t nb = (v1 + 280 - wbase + 3) / 4;
    const uint32_t *gsrc = (const uint32_t *)(V + wbase);
    for (uint32_t i = b_from + tid; i < nb; i += kMatchThreads)
      lbytes[i] = (wbase + 4 * i + 4 <= alloc_end) ? gsrc[i] : 0u;
    // "no link" (0) becomes 0xFFFF: a hop of 65535 always fails the window test, so the chain
    // step needs no separate end-of-chain compare
    const uint32_t np = (v1 - wbase + 1) / 2;
    const uint32_t *psrc = (const uint32_t *)(prevd + wbase);
    uint32_t *pdst = (uint32_t *)lprev;
    for (uint32_t i = p_from + tid; i < np; i += kMatchThreads) {
      uint32_t x = psrc[i];
      x |= (x & 0xFFFFu) ? 0u : 0xFFFFu;
      x |= (x >> 16) ? 0u : 0xFFFF0000u;
      pdst[i] = x;
    }
  };
  // what tile (q0_) needs staged
  auto set_tile = [&](uint32_t q0_) {
    q0 = q0_;
    q1 = (q0 + kTile < n) ? q0 + kTile : n;
    v0 = q0 + D;
    v1 = q1 + D;
    wbase = (v0 > kWindow + kMatchMargin ? v0 - kWindow - kMatchMargin : 0) & ~3u;
    // a tile (at most 20480 positions) straddles at most one deflate-block end (they are 32768 apart)
    const uint32_t blk0 = (q0 + 1) <= kFirstBlock ? 0u : 1u + (q0 - kFirstBlock) / kNextBlock;
    be_lo = block_end(blk0, n);
    be_hi = block_end(blk0 + 1, n);
  };
  set_tile(q0);
  stage(0, 0);
  if (tid == 0) s_next = q0;
  __syncthreads();

#if defined(LDH_MATCH_STOP) && LDH_MATCH_STOP == 1  // profiling builds only: the staging alone
  if (lbytes[tid] == 0x12345678u) mout[q0] = lprev[tid];
  return;
#endif

  // Instruction issue bounds this loop (a wave64 VALU op holds its SIMD for 4 cycles, the
  // CU's one scalar unit issues one instruction per cycle), so the chain step is written to
  // be as few instructions as possible.  Lane predicates live in wave-uniform 64-bit masks
  // (a v_cmp writes one directly; combining two costs one scalar op), and the step has no
  // per-lane "am I running" select: a lane whose candidate passed the quick reject does not
  // advance, so every later step of the round re-evaluates the same candidate with the same
  // outcome, and a lane whose chain has ended is simply not in `act`.  The long paths --
  // full compare and refill -- run once per round for all lanes that need them.
  //   act    lanes walking a chain (implies best < nice_length and best < maxlen, LD:1389)
  //   pm     lanes whose current candidate passed the quick reject in the last step
  //   pend   lanes holding a position whose result is not stored yet
  //   seed   (dictionary) lanes whose walk has taken a candidate in the last two dictionary
  //          bytes, after which LD:1394 lets it go no deeper into the dictionary
  constexpr uint32_t steps_per_round = kMatchSteps;
  // NW walks per lane (LDH_MATCH_WALKS).  With 31 registers the kernel has room for a second, independent walk in every
  // lane whose LDS round trips overlap the first one's -- measured: 22.7 ms per GiB against 21.6 with one.  The kernel
  // is bound by throughput (VALU 84 % and LDS 80 % busy), not by latency; two walks only lengthen the drain.
  constexpr int NW = LDH_MATCH_WALKS;
  struct Walk {
    unsigned long long act = 0, pm = 0, pend = 0, seed = 0;
    uint32_t q = 0, vb = 0, cb = 0, maxlen = 0, t2 = 0, dn = 0xFFFFu;
    uint32_t res = 0;  // best length << 16 | its distance: the word that is stored; 0 = no match yet
    uint32_t d1 = 0;   // candidates the walk may still evaluate after the current one (LD:1362)
    uint32_t off = 0;  // best ? best - 1 : 0
    int32_t lo = 0;    // vb - 32768: the farthest candidate inside the window (LD:1365)
    uint64_t tw = 0;
  };
  Walk w[NW];
  bool exhausted = false;
  // "is my lane in this wave-uniform mask": the mask itself as the condition (v_cmp-free inverse ballot), not an AND
  // with the lane's bit and a 64-bit compare (three vector instructions a test, a dozen a round)
  auto in_mask = [](unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); };

  // x -= 1 in the lanes of a wave-uniform mask: one subtract-with-borrow whose borrow-in is the
  // mask (the compiler's own choice is a select of 0 / 1 and a subtract)
#define LDH_DEC_MASKED(x_, m_)                                                                    \
  {                                                                                               \
    unsigned long long bo_;                                                                       \
    asm("v_subbrev_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(x_), "=s"(bo_) : "s"(m_));                \
  }
  // hop from candidate cb over link dn_ for the lanes in cond_; returns the lanes that go on
#define LDH_HOP(W_, cond_, dn_, go_)                                                              \
  {                                                                                               \
    const unsigned long long more_ = ballot64(W_.d1 != 0);                                        \
    LDH_DEC_MASKED(W_.d1, cond_);                                                                 \
    const uint32_t cn_ = W_.cb - (dn_);                                                           \
    go_ = (cond_) & more_ & ballot64((int32_t)cn_ >= W_.lo);                                      \
    if (HAS_DICT) { /* dictionary side of V (Q10) */                                              \
      const uint32_t c_ = cn_ + wbase;                                                            \
      const unsigned long long in_dict_ = ballot64(c_ < D), deep_ = ballot64(c_ + 2 < D);         \
      go_ &= ~(in_dict_ & deep_ & W_.seed);                                                       \
      W_.seed |= go_ & in_dict_ & ~deep_;                                                         \
    }                                                                                             \
    W_.cb = __builtin_amdgcn_inverse_ballot_w64(go_) ? cn_ : W_.cb;                               \
  }

  // ---- batched full compare of the candidates that passed the quick reject
  auto compare = [&](Walk &W) {
    const unsigned long long cm = W.act & W.pm;
    W.pm = 0;
    if (cm) {
      if (in_mask(cm)) {
        const uint64_t x = lds_load8(lbytes, W.cb + 3) ^ W.tw;
        uint32_t L = 3 + (x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u);
        // the tail of a long match, 8 bytes per trip: a wave-uniform loop over the lanes still equal
        // (per-lane `break`s cost some 25 scalar instructions of mask bookkeeping per trip)
        // (one ballot per compare, combined as masks: a ballot of a combined predicate goes through a 0/1 register)
        unsigned long long going = ballot64(x == 0) & ballot64(L < W.maxlen);
        while (going) {  // (bytes are staged 280 past the tile's end: reading past maxlen is safe)
          const bool on = __builtin_amdgcn_inverse_ballot_w64(going);
          const uint64_t y = lds_load8(lbytes, W.cb + L) ^ lds_load8(lbytes, W.vb + L);
          const uint32_t adv = y ? (uint32_t)__builtin_ctzll(y) >> 3 : 8u;
          L += on ? adv : 0u;
          going &= ballot64(y == 0) & ballot64(L < W.maxlen);
        }
        if (L > W.maxlen) L = W.maxlen;
        if ((L << 16) > W.res) {  // L > best (distances are below 65536); LD:1385 strict: nearest of equal lengths wins
          W.res = (L << 16) | (W.vb - W.cb);
          W.off = L - 1;
          W.t2 = lds_load2(lbytes, W.vb + W.off);
        }
      }
      // a match of maxlen cannot be beaten (the walk has no side effects, so ending it early
      // leaves the result unchanged)
      const unsigned long long open = ballot64((W.res >> 16) < nice) & ballot64((W.res >> 16) < W.maxlen);
      unsigned long long go;
      LDH_HOP(W, cm & open, W.dn, go);
      W.act = (W.act & ~cm) | go;
    }
  };
  // ---- refill idle lanes: positions [base, base + nidle) go to the idle lanes of W in lane order
  auto refill = [&](Walk &W, unsigned long long idle, uint32_t base) {
    // (The new walk state is computed by every lane -- lanes that are not idle from a clamped
    // position -- and taken over by the idle ones in one exec-masked block of plain moves: with
    // the loads inside a branch the compiler copies all twelve state registers on both sides of
    // it, per trip.  An idle lane's walk state is dead, so it is overwritten even when the lane
    // gets no position or one that is not searched (LD:1349): that one keeps best = 0 and is
    // stored like any other result at the lane's next refill.)
    const bool mine = in_mask(idle);
    if (in_mask(idle & W.pend)) mout[W.q] = W.res;
    const uint32_t my = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0));
    const unsigned long long hp = idle & ballot64(my < q1);  // idle lanes that get a position
    const bool has_pos = in_mask(hp);
    const uint32_t myc = has_pos ? my : q0;  // (keeps the LDS addresses below inside the staged range)
    const uint32_t p = myc + 1;
    const uint32_t be = p <= be_lo ? be_lo : be_hi;
    const uint32_t vbn = myc + D - wbase;
    const uint32_t d0 = lprev[vbn];
    const uint32_t cbn = vbn - d0;
    const uint32_t mln = be + 1 - p;
    const uint64_t twn = lds_load8(lbytes, vbn + 3);
    const uint32_t t2n = lds_load2(lbytes, vbn);
    const uint32_t dnn = lprev[cbn];
    if (mine) {
      W.q = myc;
      W.vb = vbn;
      W.lo = (int32_t)vbn - (int32_t)kWindow;
      W.cb = cbn;
      W.maxlen = mln > kMaxMatch ? kMaxMatch : mln;
      W.tw = twn;
      W.t2 = t2n;
      W.d1 = (uint32_t)cfg.max_chain - 1;  // Q3
      W.res = 0u;
      W.off = 0u;
      W.dn = dnn;
    }
    // LD:1349 (the last two positions of a block are not searched); chain empty (0xFFFF) / LD:1365
    const unsigned long long st = hp & ballot64(p + 2 <= be) & ballot64(d0 <= kWindow);
    W.act |= st;
    // the first candidate of a chain always passes the reject (best = 0: the bytes tested are
    // the trigram's): hand it to the full compare at once instead of freezing the lane on it
    // for a round
    W.pm |= st;
    W.pend = (W.pend & ~idle) | hp;
    if (HAS_DICT) W.seed = (W.seed & ~idle) | (st & ballot64((W.cb + wbase < D) & !(W.cb + wbase + 2 < D)));
  };

  // ---- chain steps: one LDS round trip each (link of the candidate + its reject bytes).
  // A candidate can only beat `best` if it also matches at offsets best-1 and best (the
  // trigram makes offsets 0..2 equal): two bytes reject nearly every candidate.
  auto steps = [&]() {
#pragma unroll
    for (uint32_t step = 0; step < steps_per_round; step++) {
#pragma unroll
      for (int i = 0; i < NW; i++) {
        Walk &W = w[i];
        W.dn = lprev[W.cb];
        W.pm = ballot64(lds_load2(lbytes, W.cb + W.off) == W.t2);
        unsigned long long go;
        LDH_HOP(W, W.act & ~W.pm, W.dn, go);
        W.act = (W.act & W.pm) | go;
      }
    }
  };

  // While the tile has positions left every round refills whatever is idle (about 17 lanes
  // finish per round, so a threshold saved little and its branch cost a register shuffle on
  // both sides); then the walks in flight drain.
  for (uint32_t tile = tile_first;;) {
  while (!exhausted) {
    unsigned long long idle[NW];
    uint32_t nidle[NW], total = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
      idle[i] = ~w[i].act;
      nidle[i] = (uint32_t)__popcll(idle[i]);
      total += nidle[i];
    }
    if (REFILL_MIN <= 1 || total >= (uint32_t)REFILL_MIN * NW) {
      // (one ds_add_rtn by lane 0, written out: the compiler's atomic optimiser turns a single-lane atomicAdd into its
      // generic wave reduction -- a scalar loop over the exec bits, v_writelane, two mbcnt -- some 25 instructions a round)
      uint32_t base = 0;
      if (lane == 0)
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(base)
                     : "v"((uint32_t)(uintptr_t)&s_next), "v"(total)
                     : "memory");
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int i = 0; i < NW; i++) {
        refill(w[i], idle[i], base);
        base += nidle[i];
      }
      if (base >= q1) exhausted = true;
    }
#pragma unroll
    for (int i = 0; i < NW; i++) compare(w[i]);
    steps();
  }
  // (the next tile's new bytes and links are fetched into registers now and arrive while the walks in flight drain)
  constexpr uint32_t kNewB = (kTile / 4 + THREADS - 1) / THREADS + 1, kNewP = (kTile / 2 + THREADS - 1) / THREADS + 1;
  uint32_t nwb[kNewB], nwp[kNewP];
  uint32_t kbN = 0, kpN = 0, nbN = 0, npN = 0, wbn = 0;
  const bool more = THREADS >= 1024 && tile + 1 < tile_first + tiles_per_wg && (tile + 1) * kTile < n;
  if (more) {
    const uint32_t q0n = (tile + 1) * kTile, q1n = (q0n + kTile < n) ? q0n + kTile : n;
    const uint32_t v0n = q0n + D, v1n = q1n + D;
    wbn = (v0n > kWindow + kMatchMargin ? v0n - kWindow - kMatchMargin : 0) & ~3u;
    kbN = (v1 + 280 - wbn) / 4;  // whole dwords of the bytes / links that stay (v1 = v0n)
    kpN = (v1 - wbn) / 2;
    nbN = (v1n + 280 - wbn + 3) / 4;
    npN = (v1n - wbn + 1) / 2;
    const uint32_t *gsrc = (const uint32_t *)(V + wbn);
    const uint32_t *psrc = (const uint32_t *)(prevd + wbn);
#pragma unroll
    for (uint32_t k = 0; k < kNewB; k++) {
      const uint32_t i = kbN + k * kMatchThreads + tid;
      nwb[k] = (i < nbN && wbn + 4 * i + 4 <= alloc_end) ? gsrc[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kNewP; k++) {
      const uint32_t i = kpN + k * kMatchThreads + tid;
      nwp[k] = i < npN ? psrc[i] : 0u;
    }
  }
  // The walks in flight drain -- at the run's last tile all of them; before another tile only those whose window
  // reaches below what LDS will hold then (positions more than kMatchMargin before the tile's end: rare).  The
  // others go on across the change of tiles, their LDS offsets moved along with the data (waiting for every walk
  // at every tile's end was 0.7 of the kernel's 20.7 ms per GiB).
  const uint32_t sh = more ? wbn - wbase : 0u;           // positions the window moves on: a multiple of 4
  const uint32_t must_end_below = sh ? sh + kWindow : 0u;  // walks at LDS offsets below this one
  for (;;) {
    unsigned long long any = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
      compare(w[i]);
      any |= more ? (w[i].act & ballot64(w[i].vb < must_end_below)) : w[i].act;
    }
    if (!any) break;
#if defined(LDH_MATCH_STOP) && LDH_MATCH_STOP == 2  // profiling builds only: no drain (results are wrong)
    break;
#endif
    steps();
  }
  if (!more) {
#pragma unroll
    for (int i = 0; i < NW; i++)
      if (in_mask(w[i].pend)) mout[w[i].q] = w[i].res;
    break;
  }
  // ---- the next tile of the run (a lane that is done keeps its result until its next refill, as within a tile)
  tile++;
  {
    const uint32_t v1_old = v1;
    __syncthreads();  // no walk reads LDS now
    set_tile(tile * kTile);
#pragma unroll
    for (int i = 0; i < NW; i++) {
      w[i].vb -= sh;
      w[i].cb -= sh;
      w[i].lo -= (int32_t)sh;
    }
    // what stays: bytes [wbase, v1_old + 280) and links [wbase, v1_old), through registers (the ranges overlap)
    const uint32_t kb = (v1_old + 280 - wbase) / 4, kp = (v1_old - wbase) / 2;  // whole dwords kept
    if (sh) {
      constexpr uint32_t kMaxB = (kWindow + kMatchMargin + 288 + 4 * THREADS - 1) / (4 * THREADS) + 1;
      constexpr uint32_t kMaxP = (kWindow + kMatchMargin + 2 * THREADS - 1) / (2 * THREADS) + 1;
      uint32_t rb[kMaxB], rp[kMaxP];
      uint32_t *pdst = (uint32_t *)lprev;
#pragma unroll
      for (uint32_t k = 0; k < kMaxB; k++) {
        const uint32_t i = k * kMatchThreads + tid;
        rb[k] = i < kb ? lbytes[i + sh / 4] : 0u;
      }
#pragma unroll
      for (uint32_t k = 0; k < kMaxP; k++) {
        const uint32_t i = k * kMatchThreads + tid;
        rp[k] = i < kp ? pdst[i + sh / 2] : 0u;
      }
      __syncthreads();
#pragma unroll
      for (uint32_t k = 0; k < kMaxB; k++) {
        const uint32_t i = k * kMatchThreads + tid;
        if (i < kb) lbytes[i] = rb[k];
      }
#pragma unroll
      for (uint32_t k = 0; k < kMaxP; k++) {
        const uint32_t i = k * kMatchThreads + tid;
        if (i < kp) pdst[i] = rp[k];
      }
    }
    // the new part (a dword that was kept only in part was fetched whole again)
    {
      uint32_t *pdst = (uint32_t *)lprev;
#pragma unroll
      for (uint32_t k = 0; k < kNewB; k++) {
        const uint32_t i = kbN + k * kMatchThreads + tid;
        if (i < nbN) lbytes[i] = nwb[k];
      }
#pragma unroll
      for (uint32_t k = 0; k < kNewP; k++) {
        const uint32_t i = kpN + k * kMatchThreads + tid;
        uint32_t x = nwp[k];
        x |= (x & 0xFFFFu) ? 0u : 0xFFFFu;  // "no link" becomes 0xFFFF, as in stage()
        x |= (x >> 16) ? 0u : 0xFFFF0000u;
        if (i < npN) pdst[i] = x;
      }
    }
    if (tid == 0) s_next = q0;
    exhausted = false;
    __syncthreads();
  }
  }
#undef LDH_HOP
#undef LDH_DEC_MASKED
}

// ------------------------------------------------------------------ k_parse_lazy
// LD:1318-1476, lazy branch: the state machine that turns the per-position best matches
// into tokens.  Parse state is reset per deflate block (Q6), so a block is the unit.
//
// What the machine carries from one iteration to the next is less than it looks.  With
// L(x), Dist(x) the best match at position x (independent of the parse, Q3/Q4):
//  * it arrives at iteration i with match_available set only from iteration i-1, and the
//    lookup there was not gated off (a gate `prev_len >= max_lazy_evaluation` makes that
//    iteration emit prev's match and clear match_available), so prev_len/prev_dist are
//    L(i-1)/Dist(i-1);
//  * the stale cur_len that survives a match (Q5) is <= prev_len < max_lazy or is the 0
//    of a gated lookup, so the gate of the next iteration is always open.
// Hence the token at position x is decided by L(x), Dist(x), L(x+1) alone:
//      match(x)  iff  usable(L(x), Dist(x))  and  look(x) <= L(x),
//      look(x) = L(x+1) if L(x) < max_lazy and x+1 is inside the block, else 0   (LD:1349)
// and the parse is the walk x -> x + (match ? L(x) : 1) from the block's first position.
// (tests compare this against the oracle's literal state machine.)
//
// One workgroup takes a block, tile by tile (8192 positions): L, Dist and the literal byte of
// every position are staged into LDS as one packed word (one lane per position, coalesced); then
// the walk is found by relaxation over 256 chunks of 32 positions: every lane derives the 32 step
// widths of its chunk from LDS once, walks its chunk from an assumed entry position, entries are
// re-derived as the running maximum of the exits to the left, lanes whose entry changed walk
// again, until nothing changes (chunk 0 is exact from the start and every repeat fixes at least
// one more chunk; two or three repeats are the rule).  Visited positions (a 32-bit mask per lane)
// are scanned and their tokens, rebuilt from the packed words, written out.
constexpr int kParseThreads = 256;        // 8192 positions per tile
constexpr int kParseThreadsNarrow = 128;  // 4096: batches of streams up to 4 KiB (a tile of 8192 would be half empty)
constexpr uint32_t kParseChunk = 32;

// LDS slot of tile-relative position j: one pad word per chunk, so that lanes walking one
// chunk each hit different banks
__device__ inline uint32_t parse_slot(uint32_t j) { return j + (j >> 5); }

#ifdef LDH_PARSE_STAMPS  // diagnostic builds only: shader cycles per phase of a tile (wavefront 0 of every workgroup)
__device__ unsigned long long g_parse_stamps[16];
#define LDH_PSTAMP(k_)                                             \
  {                                                                \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    ps_acc[k_] += now_ - ps_t;                                     \
    ps_t = now_;                                                   \
  }
#else
#define LDH_PSTAMP(k_)
#endif
#ifndef LDH_PARSE_WAVES
#define LDH_PARSE_WAVES 4  // four workgroups per CU (their LDS: 35 KB each)
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(LDH_PARSE_WAVES, LDH_PARSE_WAVES))) void k_parse_lazy(Batch B) {
  constexpr int kParseThreads = THREADS;
  constexpr uint32_t kParseTile = kParseThreads * kParseChunk;
  constexpr int kParseRounds = kParseTile / kParseThreads;  // positions per thread when staging
  __shared__ uint32_t s_tok[kParseTile + kParseTile / 32 + 2];
  __shared__ uint32_t s_wmax[kParseThreads / kWave];
  __shared__ uint32_t s_off[kParseThreads];
  __shared__ uint32_t s_tot[kParseThreads / kWave];
  __shared__ uint32_t s_tile_tot, s_tile_exit;
  const uint32_t unit = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t s = unit / B.max_blocks, b = unit % B.max_blocks;
  const uint32_t c = threadIdx.x, lane = c & 63, wave = c >> 6;
  if (s >= B.count) return;
  const uint32_t n = stream_len(B, s);
  if (b >= num_blocks(n)) return;
  if (block_end(b, n) < block_start(b)) {  // the one block of an empty stream
    if (c == 0) B.blocks[(size_t)s * B.max_blocks + b].ntok = 0;
    return;
  }
  const uint32_t xb0 = block_start(b) - 1, xb1 = block_end(b, n) - 1;  // 0-based, inclusive
  const uint8_t *Vd = B.vbuf + (size_t)s * (B.nvp + kVPad) + B.dlen;
  const uint32_t *m = B.bufB + (size_t)s * B.nvp;
  uint32_t *tok = B.bufA + (size_t)s * B.nvp + xb0;
  const uint32_t max_lazy = (uint32_t)level_cfg(B.level).max_lazy;

  uint32_t tile_entry = xb0, tok_base = 0;
#ifdef LDH_PARSE_STAMPS
  unsigned long long ps_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps_t = __builtin_amdgcn_s_memtime();
  uint32_t ps_tiles = 0, ps_reps = 0;
#endif
  // A thread stages four consecutive positions at a time: one 16-byte load of the matches and one 4-byte load of the
  // bytes, neither aligned beyond what the arrays are (a sixth of the load instructions of one position per thread and
  // load; reading up to three entries past the block's end stays inside the arrays' padding).  The loads of a tile are
  // issued before the tile before it stores its tokens: memory operations complete in the order they were issued, so
  // loads behind those scattered stores would wait for every one of them.
  typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef uint32_t u32_a1 __attribute__((aligned(1)));
  constexpr int kGroups = kParseRounds / 4;  // groups of 4 * THREADS positions
  u32x4_a4 mm[kGroups];
  uint32_t bb[kGroups], ahead = 0;  // ahead: the match at the first position of the tile after (only its length is looked at)
  auto fetch = [&](uint32_t x0_) {
    {
      const uint32_t xe = x0_ + kParseTile;
      ahead = m[xe <= xb1 ? xe : xb1];
    }
#pragma unroll
    for (int r = 0; r < kGroups; r++) {
      const uint32_t x = x0_ + r * 4 * kParseThreads + 4 * c;
      const uint32_t xc = x <= xb1 ? x : xb1;
      mm[r] = *(const u32x4_a4 *)(m + xc);
      bb[r] = *(const u32_a1 *)(Vd + xc);
    }
  };
  fetch(xb0);
  for (uint32_t x0 = xb0; x0 <= xb1; x0 += kParseTile) {
    LDH_PSTAMP(5)
    // ---- stage the tile: one word per position = literal byte | (dist - 1) << 8 | length << 23
    // (8 + 15 + 9 bits), so that step widths and tokens can both be derived from LDS alone.  Half a
    // tile at a time: all loads of a half are issued before any of them is waited for; addresses
    // are clamped rather than branched around.
    // ---- (the tile's words were loaded before the tile before it wrote its tokens out: see `fetch` below)
#pragma unroll
    for (int r = 0; r < kGroups; r++) {
      const uint32_t j0 = r * 4 * kParseThreads + 4 * c;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t mk = mm[r][k];
        const uint32_t w = ((bb[r] >> (8 * k)) & 0xFFu) | (((mk - 1u) & 0x7FFFu) << 8) | ((mk >> 16) << 23);
        s_tok[parse_slot(j0 + k)] = x0 + j0 + k <= xb1 ? w : 0u;
      }
    }
    if (c == 0) s_tok[parse_slot(kParseTile)] = x0 + kParseTile <= xb1 ? (ahead >> 16) << 23 : 0u;
    __syncthreads();
    LDH_PSTAMP(0)

    // ---- the walk, by relaxation over chunks.  A lane keeps the 32 step widths of its chunk
    // in registers, so a (re-)walk is straight-line code with no memory access.
    // step(x): match iff usable(L, dist) and look <= L, look = L(x+1) if L < max_lazy and x+1 is inside
    // the block, else 0 (see above)
    const uint32_t cs = x0 + c * kParseChunk;
    const uint32_t clen = cs > xb1 ? 0u : (xb1 + 1 - cs < kParseChunk ? xb1 + 1 - cs : kParseChunk);
    uint32_t is_match = 0;  // bit k: position cs + k, if the walk comes by, emits a match
    {
      uint32_t w = s_tok[c * (kParseChunk + 1)];
#pragma unroll
      for (int k = 0; k < (int)kParseChunk; k++) {
        const uint32_t wn = s_tok[c * (kParseChunk + 1) + k + 1 + ((k + 1) >> 5)];  // = parse_slot(c * 32 + k + 1)
        const uint32_t pl = w >> 23, pd1 = (w >> 8) & 0x7FFFu;
        const uint32_t look = ((pl < max_lazy) & (cs + k + 1 <= xb1)) ? wn >> 23 : 0u;
        const bool mt = ((pl > 3) | ((pl == 3) & (pd1 < 4095u))) & (look <= pl);  // LD:1402-1404
        is_match |= mt ? 1u << k : 0u;  // (positions beyond the block's end were staged as 0: no bit)
        w = wn;
      }
    }
    uint32_t entry = c == 0 ? tile_entry : cs, exit_x = 0, mask = 0;
    bool redo = true;
    LDH_PSTAMP(1)
    for (;;) {
#ifdef LDH_PARSE_STAMPS
      ps_reps++;
#endif
      if (redo) {
        // (entry >= cs always: it is an exit of a chunk to the left, or the assumed cs)
        // The walk hops from match to match: between two of them every position is a literal and is visited, so a
        // hop is "the next bit of is_match at or above k", a run of mask bits, and the match's length from LDS --
        // three or four hops a chunk on text, where stepping through all 32 positions (their widths in 32 registers)
        // was 160 dependent instructions a walk and a third of the kernel.
        uint32_t k = entry - cs;
        mask = 0;
        exit_x = entry;
        while (k < clen) {
          const uint32_t mm = is_match >> k;
          if (mm == 0) {  // literals to the chunk's end
            mask |= (clen >= 32 ? ~0u : (1u << clen) - 1u) & ~((1u << k) - 1u);
            exit_x = cs + clen;
            break;
          }
          const uint32_t j = k + (uint32_t)__builtin_ctz(mm);  // the next match the walk meets
          mask |= ((2u << j) - 1u) & ~((1u << k) - 1u);           // positions k .. j
          const uint32_t nx = j + (s_tok[c * (kParseChunk + 1) + j] >> 23);
          exit_x = cs + nx;  // (stands when the match leaves the chunk)
          k = nx;
        }
      }
      // entry of chunk c = the largest exit to its left (exits of the true walk never
      // decrease).  A chunk the walk only passes through (or an empty one) contributes
      // nothing of its own, so that a wrong value to its left is gone the moment its source
      // is corrected instead of being handed on one chunk per repeat.
      const uint32_t inc = wave_inclusive_max(mask ? exit_x : 0u);
      if (lane == 63) s_wmax[wave] = inc;
      __syncthreads();
      uint32_t e = tile_entry;
      for (uint32_t w = 0; w < wave; w++) e = s_wmax[w] > e ? s_wmax[w] : e;
      // (wave_shr:1 -- the lane to the left across the rows of 16; lane 0 gets 0)
      const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xF, 0xF, false);
      if (left > e) e = left;
      redo = e != entry;
      entry = e;
      if (!__syncthreads_or(redo)) break;
    }

    LDH_PSTAMP(2)
    if (x0 + kParseTile <= xb1) fetch(x0 + kParseTile);  // (uniform)
    // ---- emit the visited positions' tokens
    const uint32_t cnt = __popc(mask);
    scan256_exclusive(cnt, c, s_off, s_tot);
    LDH_PSTAMP(3)
    uint32_t o = tok_base + s_off[c];
    while (mask) {
      const uint32_t bit = (uint32_t)__builtin_ctz(mask);
      mask &= mask - 1;
      const uint32_t j = cs + bit - x0;
      const uint32_t w = s_tok[parse_slot(j)];
      tok[o++] = ((is_match >> bit) & 1u) ? ((1u << 31) | (((w >> 23) - 3u) << 16) | ((w >> 8) & 0x7FFFu)) : (w & 0xFFu);
    }
    if (c == kParseThreads - 1) {
      s_tile_tot = s_off[c] + cnt;
      s_tile_exit = exit_x > entry ? exit_x : entry;
    }
    __syncthreads();
    LDH_PSTAMP(4)
#ifdef LDH_PARSE_STAMPS
    ps_tiles++;
#endif
    tok_base += s_tile_tot;
    tile_entry = s_tile_exit;
  }
#ifdef LDH_PARSE_STAMPS
  if (c == 0) {
    for (int i = 0; i < 8; i++) atomicAdd(&g_parse_stamps[i], ps_acc[i]);
    atomicAdd(&g_parse_stamps[14], (unsigned long long)ps_reps);
    atomicAdd(&g_parse_stamps[15], (unsigned long long)ps_tiles);
  }
#endif
  if (c == 0) B.blocks[(size_t)s * B.max_blocks + b].ntok = tok_base;
}

// ------------------------------------------------------------------ k_lz_greedy
// Levels 1-3 (LD:1318-1476, non-lazy branch).  Positions inside a match longer than
// max_insert_length never enter the hash chains (Q4, LD:1249/1454), so the candidate set of
// a position depends on the parse before it: the walk over a stream is serial.  One
// wavefront takes one stream and spends its lanes on the candidates of the current position:
// in the array sorted by (trigram, position) the chain of position v is the entries before
// rank[v] (each with the first four bytes of its position), so 64 candidates are fetched with
// one coalesced load, filtered with ballots
// (same trigram, window LD:1365, dictionary rule Q10/LD:1394, "was inserted" from a bitmap of
// the last 40960 positions in LDS), the nearest max_chain survivors are compared in
// parallel, and nice_length (LD:1389) cuts the list where the serial walk would have stopped.
__device__ inline uint32_t gld4(const uint8_t *V, uint32_t off) {  // V is 16-byte aligned and padded
  const uint32_t *w = (const uint32_t *)(V + (off & ~3u));
  return __builtin_amdgcn_alignbyte(w[1], w[0], off & 3);
}

// One wavefront per stream.  A 128 KiB stream has some 30000 tokens and a token costs the dependent round trips to
// global memory it makes (the sorted arrays of 8192 streams are 17 GB: every fetch is a miss), so a token is built to
// need ONE round trip in the usual case:
//  * the rank and first bytes of the next 32 positions are fetched while the current position is being worked on,
//    so the next position (x + token length) starts without a round trip of its own;
//  * an entry of the sorted array carries the first twelve bytes of its position, so a candidate's length is known
//    from the fetch that found it unless it reaches twelve (then the whole wave compares the next 256 bytes of that
//    one candidate at once: one more round trip, whatever the length);
//  * level 1 fetches 16 candidates per trip (it wants 4 inserted ones), level 2 32, level 3 64.
// Measured alternatives (all byte-identical, none faster; ms per GiB at level 1 against 79 for this one): four
// streams per wavefront, sixteen lanes each, to quarter the instructions per token (89: the 5 KB "was inserted" ring
// per stream leaves eight such wavefronts per CU, too few to hide the round trips), two streams of 32 lanes (88),
// fetches of whole 128-byte lines of the sorted array (93: fewer candidates per trip, more trips).  The walk is bound by
// the loaded latency of its scattered reads (169 GB of 128-byte misses per GiB of input, 1.9 TB/s).
// What counts besides is how many streams a CU keeps in flight: a ring of 40960 positions (window + longest match +
// slack; 5 KB) and 64 VGPRs let 8 wavefronts per SIMD be resident.
constexpr uint32_t kInsWords = 1280;
constexpr uint32_t kGreedyAhead = 32;  // positions fetched ahead
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lz_greedy(Batch B) {
  __shared__ uint32_t s_ins[kInsWords];  // bit (q mod 40960): stream position q is in a hash chain (LD:1347)
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  if (s >= B.count) return;
  const uint32_t n = stream_len(B, s), D = B.dlen;
  const uint8_t *V = B.vbuf + (size_t)s * (B.nvp + kVPad);
  const uint4 *sorted = (const uint4 *)B.bufC + (size_t)s * B.nvp;  // {position, its first twelve bytes}
  const uint32_t *rank = B.bufB + (size_t)s * B.nvp;
  const LevelCfg cfg = level_cfg(B.level);
  const uint32_t nice = (uint32_t)cfg.nice, max_chain = (uint32_t)cfg.max_chain;
  const uint32_t max_insert = (uint32_t)cfg.max_lazy;  // max_insert_length for these levels
  const uint32_t W = max_chain <= 4 ? 16u : (max_chain <= 8 ? 32u : 64u);  // candidates fetched per trip
  const uint32_t nblk = num_blocks(n);
  const unsigned long long below = (1ull << lane) - 1ull;
  for (uint32_t i = lane; i < kInsWords; i += kWave) s_ins[i] = 0;
  __syncthreads();

  for (uint32_t b = 0; b < nblk; b++) {
    const uint32_t bs = block_start(b), be = block_end(b, n);  // 1-based, inclusive
    uint32_t *tok = B.bufA + (size_t)s * B.nvp + (bs - 1);
    uint32_t ntok = 0, x = bs - 1;  // x: 0-based stream position
    uint32_t cur_r = __builtin_amdgcn_readfirstlane(rank[x + D]);
    uint32_t my0 = __builtin_amdgcn_readfirstlane(gld4(V, x + D));
    uint32_t my1 = __builtin_amdgcn_readfirstlane(gld4(V, x + D + 4));
    uint32_t my2 = __builtin_amdgcn_readfirstlane(gld4(V, x + D + 8));
    while (x + 1 <= be) {
      const uint32_t v = x + D;
      // lanes 32.. repeat the addresses of lanes 0..31: no further memory transactions
      const uint32_t pv = v + 1 + (lane & (kGreedyAhead - 1)) < D + n ? v + 1 + (lane & (kGreedyAhead - 1)) : D + n - 1;
      const uint32_t pre_r = rank[pv], pre_w = gld4(V, pv);
      uint32_t best = 0, bdist = 0;
      if (x + 3 <= be) {  // LD:1349
        uint32_t maxlen = be - x;
        if (maxlen > kMaxMatch) maxlen = kMaxMatch;
        const uint32_t key = my0 & 0xFFFFFFu;
        const uint32_t r = cur_r;
        uint32_t taken = 0;
        bool seen_seed = false;
        for (uint32_t base = 0;; base += W) {
          const bool have = (lane < W) & (r >= 1 + base + lane);
          uint4 ce = make_uint4(0u, ~my0, 0u, 0u);
          if (have) ce = sorted[r - 1 - base - lane];
          const uint32_t c = ce.x;
          const bool ok = have && (ce.y & 0xFFFFFFu) == key && v - c <= kWindow;  // (another group's c may exceed v)
          // the chain is the run of ok lanes from lane 0
          const unsigned long long notok = ~ballot64(ok);
          const uint32_t first_bad = notok ? (uint32_t)__builtin_ctzll(notok) : 64u;  // (<= W)
          const bool in_chain = lane < first_bad;
          const bool in_dict = c < D, deep = in_dict && c + 2 < D;
          unsigned long long seedm = 0;
          uint32_t first_brk = 64;
          if (D) {  // (uniform) the dictionary rule only exists with a dictionary
            seedm = ballot64(in_chain && in_dict && !deep);
            const bool seen_before = seen_seed || (seedm & below) != 0;
            const unsigned long long brk = ballot64(in_chain && deep && seen_before);  // LD:1394
            first_brk = brk ? (uint32_t)__builtin_ctzll(brk) : 64u;
          }
          const uint32_t q = c - D;
          const bool inserted = in_dict || ((s_ins[(q >> 5) % kInsWords] >> (q & 31)) & 1);
          const bool cand = in_chain && lane < first_brk && inserted;
          const unsigned long long candm = ballot64(cand);
          const bool sel = cand && taken + (uint32_t)__popcll(candm & below) < max_chain;  // LD:1362
          // length from the entry's twelve bytes (the trigram is known to be equal)
          uint32_t L = 0;
          if (sel) {
            const uint32_t x0 = ce.y ^ my0, x1 = ce.z ^ my1, x2 = ce.w ^ my2;
            L = x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : (x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : (x2 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : 12u));
            if (L > maxlen) L = maxlen;  // LD:1356-1359: a match never crosses the block's end
          }
          const unsigned long long selm = ballot64(sel);
          // candidates that reached twelve bytes get their true length from the whole wave, nearest first, until one
          // reaches nice_length (the walk would have stopped there, LD:1389: the ones behind it do not matter)
          unsigned long long lm = ballot64(sel && L >= 12 && maxlen > 12);
          if (nice <= 12) {  // (uniform) the first candidate of 8 (level 1) ends the walk whatever its true length
            const unsigned long long n12 = ballot64(sel && L >= nice);
            if (n12) lm &= (n12 & (0ull - n12)) | ((n12 & (0ull - n12)) - 1ull);
          }
          while (lm) {
            const uint32_t j = (uint32_t)__builtin_ctzll(lm);
            lm &= lm - 1;
            const uint32_t cj = __builtin_amdgcn_readlane(c, j);
            const uint32_t off = 12 + 4 * lane;  // (12 + 4 * 63 > 258: some lane always ends the comparison)
            uint32_t y = 1;
            if (off < maxlen) y = gld4(V, cj + off) ^ gld4(V, v + off);
            const unsigned long long mism = ballot64(y != 0);
            const uint32_t f = (uint32_t)__builtin_ctzll(mism);
            const uint32_t yf = __builtin_amdgcn_readlane(y, f);
            uint32_t Lj = 12 + 4 * f;
            if (Lj < maxlen) Lj += (uint32_t)__builtin_ctz(yf) >> 3;  // (a lane at or beyond maxlen holds y = 1: + 0)
            if (Lj > maxlen) Lj = maxlen;
            L = lane == j ? Lj : L;
            if (Lj >= nice) lm = 0;
          }
          // nice_length: the walk stops at the nearest candidate that reaches it (LD:1389)
          const unsigned long long nicem = ballot64(sel && L >= nice);
          const uint32_t first_nice = nicem ? (uint32_t)__builtin_ctzll(nicem) : 64u;
          // longest of the candidates up to there, nearest among equals (LD:1385)
          unsigned long long m = selm & (first_nice >= 63 ? ~0ull : ((2ull << first_nice) - 1ull));
          if (m) {
            if (__popcll(m) > 6) {  // narrow the set down bit by bit of L, most significant first
#pragma unroll
              for (int bit = 8; bit >= 0; bit--) {
                const unsigned long long t = m & ballot64(((L >> bit) & 1) != 0);
                if (t) m = t;
              }
            }
            do {  // (a handful of lanes: scalar scan, nearest first)
              const uint32_t j = (uint32_t)__builtin_ctzll(m);
              m &= m - 1;
              const uint32_t Lj = __builtin_amdgcn_readlane(L, j);
              if (Lj > best) {
                best = Lj;
                bdist = v - __builtin_amdgcn_readlane(c, j);
              }
            } while (m);
          }
          taken += (uint32_t)__popcll(selm);
          seen_seed = seen_seed || seedm != 0;
          if (nicem || first_bad < W || first_brk < 64 || taken >= max_chain) break;
        }
      }
      // ---- the token, and which of its positions enter the chains
      uint32_t len = 1;
      bool interior_on = true;
      if (best > 3 || (best == 3 && bdist < 4096)) {
        if (lane == 0) tok[ntok] = tok_match(best, bdist);
        len = best;
        interior_on = best <= max_insert;  // LD:1454
      } else {
        if (lane == 0) tok[ntok] = my0 & 0xFFu;
      }
      ntok++;
      {
        const uint32_t first = x, last = x + len - 1, wi = (first >> 5) + lane;
        if (wi <= (last >> 5)) {
          const uint32_t lo = first > wi * 32 ? first - wi * 32 : 0u;
          const uint32_t hi = last < wi * 32 + 31 ? last - wi * 32 : 31u;
          const uint32_t mask = (hi == 31 ? ~0u : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
          const uint32_t val = interior_on ? mask : (lane == 0 ? 1u << (first & 31) : 0u);
          const uint32_t slot = wi % kInsWords;
          s_ins[slot] = (s_ins[slot] & ~mask) | val;
        }
      }
      __syncthreads();
      x += len;
      if (len + 8 <= kGreedyAhead) {  // the new position's rank and twelve bytes are among what was fetched ahead
        cur_r = __builtin_amdgcn_readlane(pre_r, len - 1);
        my0 = __builtin_amdgcn_readlane(pre_w, len - 1);
        my1 = __builtin_amdgcn_readlane(pre_w, len + 3);
        my2 = __builtin_amdgcn_readlane(pre_w, len + 7);
      } else if (x < n) {
        cur_r = __builtin_amdgcn_readfirstlane(rank[x + D]);
        my0 = __builtin_amdgcn_readfirstlane(gld4(V, x + D));
        my1 = __builtin_amdgcn_readfirstlane(gld4(V, x + D + 4));
        my2 = __builtin_amdgcn_readfirstlane(gld4(V, x + D + 8));
      }
    }
    if (lane == 0) B.blocks[(size_t)s * B.max_blocks + b].ntok = ntok;
  }
}

// ------------------------------------------------------------------ levels 1-2, second form: k_cand_lists + k_lz_greedy2
//   k_lz_greedy above pays one dependent round trip to global memory per token and is bound by
// the scalar unit besides (113 VALU + 135 SALU per token, one token at a time).  The walk over a stream must stay
// serial (Q4: which positions are in the chains depends on the tokens before), but neither the candidates of a
// position nor their match lengths depend on the parse -- only which of them count as inserted does.  So:
//   k_cand_lists   (all positions in parallel, window and links in LDS like k_match_lazy) writes for every position
//                  its K nearest same-trigram predecessors inside the window, nearest first: distance - 1 (15 bits) and
//                  the match length, capped at 18, as a nibble.  K = 8 at level 1, 16 at level 2 (the walk wants 4 / 8
//                  inserted candidates; on text 86 % / 85 % of the searched positions are decided inside the list).
//   k_lz_greedy2   one wavefront per stream takes 64 consecutive positions a round.  Phase 1, all lanes: lane i reads
//                  the list of position c0 + i (consecutive lanes, consecutive memory: a stream, not a gather) and
//                  runs LD:1362-1398 on it against the "was inserted" bitmap (a ring of 32768 bits in LDS); everything
//                  before c0 is final, so the result is final too unless a candidate lies inside the round's own 64
//                  positions (then the lane is marked uncertain) or the list ends with the chain still going on
//                  (needs more).  Phase 2, wave-uniform: the walk x -> x + token length over the 64 positions picks
//                  the real tokens; an uncertain lane that the walk comes by is evaluated again (by then everything
//                  before it is final), a lane that needs more goes on through the list of its last candidate (its
//                  candidates' lengths then come from byte compares, 16 candidates a trip, as in k_lz_greedy).
//                  Tokens are gathered in a register and stored 64 at a time.
// Per 64 positions the serial part is the ~15 real tokens' bookkeeping; the candidate evaluation, done for all 64
// positions whether the parse comes by or not, is what the lanes are for.  No dictionary (batches with one, and
// level 3, whose walk wants 32 inserted candidates, keep k_lz_greedy).
constexpr uint32_t kCandTile = 16384;
constexpr uint32_t kCandByteCap = kWindow + kCandTile + 288;
constexpr uint32_t kCandPrevCap = kWindow + kCandTile + 8;
constexpr uint32_t kCandLdsBytes = kCandByteCap + 2 * kCandPrevCap;
constexpr int kCandThreads = 1024, kCandThreadsNarrow = 256;
constexpr uint32_t kCandLenCap = 18;  // lengths are stored as min(L, 18) - 3

// per stream: [nvp * 2K bytes of distances][nvp * K/2 bytes of length nibbles] in bufC
__host__ __device__ inline size_t cand_stream_bytes(uint32_t nvp, int K) { return (size_t)nvp * (size_t)(2 * K + K / 2); }

template <int K, int THREADS>
__global__ __launch_bounds__(THREADS) void k_cand_lists(Batch B, uint32_t tiles_per_stream, uint32_t lds_byte_cap) {
  extern __shared__ uint32_t smem[];
  uint32_t *lbytes = smem;
  uint16_t *lprev = (uint16_t *)(smem + lds_byte_cap / 4);
  const uint32_t work = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t s = work / tiles_per_stream, tile = work % tiles_per_stream;
  const uint32_t tid = threadIdx.x;
  const uint32_t n = stream_len(B, s);
  const uint32_t q0 = tile * kCandTile;
  if (q0 >= n) return;
  const uint32_t q1 = q0 + kCandTile < n ? q0 + kCandTile : n;
  const uint8_t *V = B.vbuf + (size_t)s * (B.nvp + kVPad);
  const uint16_t *prevd = B.prevd + (size_t)s * B.nvp;
  uint8_t *cbase = (uint8_t *)B.bufC + (size_t)s * cand_stream_bytes(B.nvp, K);
  uint32_t *dout = (uint32_t *)cbase;                               // K/2 dwords per position
  uint32_t *nout = (uint32_t *)(cbase + (size_t)B.nvp * 2 * K);     // K/8 dwords per position
  const uint32_t alloc_end = n + kVPad;
  const uint32_t wbase = (q0 > kWindow ? q0 - kWindow : 0u) & ~3u;
  {  // stage bytes [wbase, q1 + 280) and links [wbase, q1): as k_match_lazy does ("no link" becomes a hop of 65535)
    const uint32_t nb = (q1 + 280 - wbase + 3) / 4;
    const uint32_t *gsrc = (const uint32_t *)(V + wbase);
    for (uint32_t i = tid; i < nb; i += THREADS) lbytes[i] = (wbase + 4 * i + 4 <= alloc_end) ? gsrc[i] : 0u;
    const uint32_t np = (q1 - wbase + 1) / 2;
    const uint32_t *psrc = (const uint32_t *)(prevd + wbase);
    uint32_t *pdst = (uint32_t *)lprev;
    for (uint32_t i = tid; i < np; i += THREADS) {
      uint32_t x = psrc[i];
      x |= (x & 0xFFFFu) ? 0u : 0xFFFFu;
      x |= (x >> 16) ? 0u : 0xFFFF0000u;
      pdst[i] = x;
    }
  }
  __syncthreads();
  const uint32_t blk0 = (q0 + 1) <= kFirstBlock ? 0u : 1u + (q0 - kFirstBlock) / kNextBlock;
  const uint32_t be_lo = block_end(blk0, n), be_hi = block_end(blk0 + 1, n);
  for (uint32_t q = q0 + tid; q < q1; q += THREADS) {
    const uint32_t p = q + 1, be = p <= be_lo ? be_lo : be_hi;
    const uint32_t vb = q - wbase;
    const int32_t lo = (int32_t)vb - (int32_t)kWindow;  // LD:1365
    uint32_t dd[K / 2];
    uint32_t nb[K / 8];
#pragma unroll
    for (int i = 0; i < K / 2; i++) dd[i] = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < K / 8; i++) nb[i] = 0;
    // (every position gets its list, also the last two of a block, which LD:1349 does not search: k_lz_greedy2 goes on
    // through the lists of candidates when a chain is longer than K, and those positions are chain members like any)
    bool run = true;
    uint32_t maxlen = be + 1 - p;
    maxlen = maxlen > kMaxMatch ? kMaxMatch : (maxlen < 3 ? 3u : maxlen);
    const uint64_t tw0 = lds_load8(lbytes, vb + 3), tw1 = lds_load8(lbytes, vb + 11);
    uint32_t c = vb;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t d = lprev[c];
      const uint32_t cn = c - d;
      run = run & ((int32_t)cn >= lo);
      c = run ? cn : c;
      const uint64_t x = lds_load8(lbytes, c + 3) ^ tw0;
      uint32_t L = 3 + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u);
      if (__builtin_amdgcn_ballot_w64(run & (x == 0))) {  // (uniform: the second eight bytes are rarely needed)
        const uint64_t y = lds_load8(lbytes, c + 11) ^ tw1;
        L = x ? L : 11 + (y ? (uint32_t)__builtin_ctzll(y) >> 3 : 8u);
      }
      L = L > maxlen ? maxlen : L;
      L = L > kCandLenCap ? kCandLenCap : L;
      const uint32_t ent = run ? vb - c - 1 : 0xFFFFu;
      dd[k >> 1] = (k & 1) ? ((dd[k >> 1] & 0xFFFFu) | (ent << 16)) : ((dd[k >> 1] & 0xFFFF0000u) | ent);
      nb[k >> 3] |= run ? (L - 3) << (4 * (k & 7)) : 0u;
    }
    // bit 15 of the last entry: the chain goes on inside the window
    {
      const uint32_t d = lprev[c];
      const bool more = run & ((int32_t)(c - d) >= lo);
      dd[K / 2 - 1] |= more ? 0x80000000u : 0u;
    }
    if (K == 8) {
      *(uint4 *)(dout + (size_t)q * 4) = make_uint4(dd[0], dd[1], dd[2], dd[3]);
      nout[q] = nb[0];
    } else {
      *(uint4 *)(dout + (size_t)q * 8) = make_uint4(dd[0], dd[1], dd[2], dd[3]);
      *(uint4 *)(dout + (size_t)q * 8 + 4) = make_uint4(dd[4 % (K / 2)], dd[5 % (K / 2)], dd[6 % (K / 2)], dd[7 % (K / 2)]);
      *(uint2 *)(nout + (size_t)q * 2) = make_uint2(nb[0], nb[1 % (K / 8)]);
    }
  }
}

// The bitmap is written and read by the lanes of ONE wavefront, and LDS operations of a wavefront execute in order: all
// that is needed between a write and the reads behind it is that the compiler keeps them in that order.  (A workgroup
// fence -- or __syncthreads(), as in k_lz_greedy -- also waits for every global store and load in flight: with a token
// store before it, that was a round trip to memory per token, 23 of this kernel's first 47 ms per GiB.)
__device__ inline void lds_order() { asm volatile("" ::: "memory"); }

template <int K>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lz_greedy2(Batch B) {
  __shared__ uint32_t s_ins[1024];  // bit (q mod 32768): stream position q is in a hash chain (LD:1347)
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  if (s >= B.count) return;
  const uint32_t n = stream_len(B, s);
  const uint8_t *V = B.vbuf + (size_t)s * (B.nvp + kVPad);
  const uint8_t *cbase = (const uint8_t *)B.bufC + (size_t)s * cand_stream_bytes(B.nvp, K);
  const uint32_t *dg = (const uint32_t *)cbase;
  const uint32_t *ng = (const uint32_t *)(cbase + (size_t)B.nvp * 2 * K);
  const uint16_t *dg16 = (const uint16_t *)cbase;
  // (the level's parameters as constants -- the list length K stands for the level: 8 = level 1, 16 = level 2 -- so that
  // tests like `j < max_insert` are decided at compile time instead of by a scalar branch each)
  constexpr uint32_t nice = K == 8 ? 8u : 18u, max_chain = K == 8 ? 4u : 8u;  // level_cfg(1 / 2)
  constexpr uint32_t max_insert = K == 8 ? 4u : 5u;                         // max_insert_length (LD:1249-1250)
  const uint32_t nblk = num_blocks(n);
  const unsigned long long below = (1ull << lane) - 1ull;
  for (uint32_t i = lane; i < 1024; i += kWave) s_ins[i] = 0;
  __syncthreads();
  enum : uint32_t { ST_DONE = 0, ST_RUN = 1, ST_UNCERTAIN = 2, ST_MORE = 3 };

  for (uint32_t b = 0; b < nblk; b++) {
    const uint32_t bs = block_start(b), be = block_end(b, n);  // 1-based, inclusive
    uint32_t *tok = B.bufA + (size_t)s * B.nvp + (bs - 1);
    uint32_t ntok = 0, x = bs - 1;  // x: 0-based stream position
    while (x + 1 <= be) {
      const uint32_t c0 = x;
      const uint32_t p = c0 + lane;  // this lane's position (0-based)
      const bool valid = p + 1 <= be;
      // ---- phase 1: the list of position p against the bitmap
      uint32_t dd[K / 2], nb[K / 8];
      {
        const uint32_t pc = valid ? p : c0;
        if (K == 8) {
          const uint4 a = *(const uint4 *)(dg + (size_t)pc * 4);
          dd[0] = a.x, dd[1] = a.y, dd[2] = a.z, dd[3] = a.w;
          nb[0] = ng[pc];
        } else {
          const uint4 a = *(const uint4 *)(dg + (size_t)pc * 8), c = *(const uint4 *)(dg + (size_t)pc * 8 + 4);
          const uint2 e = *(const uint2 *)(ng + (size_t)pc * 2);
          dd[0] = a.x, dd[1] = a.y, dd[2] = a.z, dd[3] = a.w;
          dd[4 % (K / 2)] = c.x, dd[5 % (K / 2)] = c.y, dd[6 % (K / 2)] = c.z, dd[7 % (K / 2)] = c.w;
          nb[0] = e.x, nb[1 % (K / 8)] = e.y;
        }
      }
      const uint32_t litv = V[valid ? p : c0];
      // LD:1362-1398 over the list, for the lanes in `act`; candidates at or beyond `final_below` are not final yet
      // (The walk's state is two lane predicates -- still running, met a candidate that is not final -- kept as wave
      // masks; the bitmap word is read whether or not the lane still runs: its index is masked into the ring.)
      uint32_t best, bd, st, depth, lastd;
      auto eval = [&](bool act, uint32_t final_below) {
        best = 0, bd = 0, depth = max_chain;
        bool run = act, unc = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
          const uint32_t e = (dd[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
          const bool end = e == 0xFFFFu;  // (an entry is never 0xFFFF otherwise: a distance of 32768 has no successor)
          const uint32_t dist = (e & 0x7FFFu) + 1;
          const uint32_t q = p - dist;  // (no dictionary: q >= 0 for every entry phase A wrote)
          const bool live = run & !end;
          const bool early = live & (q >= final_below);
          const bool look = live & !early;
          const uint32_t w = s_ins[(q >> 5) & 1023u];
          const bool ins = look & (((w >> (q & 31)) & 1u) != 0);
          const uint32_t L = ((nb[k >> 3] >> (4 * (k & 7))) & 15u) + 3;
          const bool better = ins & (L > best);  // LD:1385
          bd = better ? dist : bd;
          best = better ? L : best;
          depth -= ins ? 1u : 0u;
          unc = unc | early;
          run = look & !(ins & ((best >= nice) | (depth == 0)));
        }
        // the list is used up and the walk still runs: more candidates only if the chain goes on (bit 15 of the last
        // entry); it then goes on from the list's last candidate
        const bool more = (dd[K / 2 - 1] >> 31) != 0;
        lastd = ((dd[K / 2 - 1] >> 16) & 0x7FFFu) + 1;
        st = unc ? (uint32_t)ST_UNCERTAIN : ((run & more) ? (uint32_t)ST_MORE : (uint32_t)ST_DONE);
      };
      eval(valid && p + 3 <= be /* LD:1349 */, c0);
      const uint32_t best1 = best, bd1 = bd, st1 = st, depth1 = depth, lastd1 = lastd;
      // ---- phase 2: the walk over these 64 positions.  The scalar unit only hops from token to token over the lanes
      // whose result is final (one v_readlane + add a token); what each of those tokens needs done -- forming it,
      // storing it, its bits of the bitmap -- is done by the lanes, for all of them at once.  A lane whose result is not
      // final stops the hop: it is resolved on its own (wave-uniform code), then the hopping goes on behind it.
      const uint32_t limit = be - c0 < kWave ? be - c0 : kWave;  // lanes with a position inside the block
      const bool m1 = (best1 > 3) | ((best1 == 3) & (bd1 < 4096));  // LD:1402-1404
      const uint32_t len1 = m1 ? best1 : 1u;
      uint32_t maxlen_l = be - p;  // (this lane's position)
      maxlen_l = maxlen_l > kMaxMatch ? kMaxMatch : maxlen_l;
      // not final: uncertain, needs more candidates, or a length of "18 or more"
      const unsigned long long special =
          ballot64(valid & ((st1 != ST_DONE) | (m1 & (best1 >= kCandLenCap) & (maxlen_l > kCandLenCap))));
      const uint32_t tok1 = m1 ? tok_match(best1, bd1) : litv;
      // What a round decides is kept in registers until its end: visF / visR the lanes whose position starts a token
      // (final lanes the hop came by / lanes resolved on their own), tokv the token of each, R the bitmap bits of the
      // resolved tokens (relative to c0).  Tokens and bitmap words are written once, behind the walk; a resolution that
      // must know whether a position of this round is in the chains reads the bits from these masks instead of LDS.
      // (Storing the tokens and rewriting the bitmap after every stretch of final lanes was some 65 instructions a
      // stretch, six stretches a round.)
      // Sj: final lanes whose token is a match short enough for its interior to be inserted and longer than j
      unsigned long long Sj[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (uint32_t j = 1; j < 6; j++)
        if (j < max_insert) Sj[j - 1] = ballot64(m1 & (len1 <= max_insert)) & ballot64(len1 > j);  // (uniform test)
      auto final_bits_lo = [&](unsigned long long vf) {
        unsigned long long m = vf;
#pragma unroll
        for (uint32_t j = 1; j < 6; j++)
          if (j < max_insert) m |= (vf & Sj[j - 1]) << j;
        return m;
      };
      unsigned long long visF = 0, visR = 0, Rlo = 0, Rhi = 0;
      uint32_t tokv = tok1;
      uint32_t cur = 0;
      while (cur < limit) {
        {
          // while (k < limit && !bit(special, k)) { visF |= bit k; k += len1 of lane k; } -- seven instructions a token
          // written out (the compiler's version shifts and masks 64-bit values: eleven)
          uint32_t k = cur, hop_t;
          asm volatile(
              "s_cmp_lt_u32 %[k], %[lim]\n\t"
              "s_cbranch_scc0 2f\n"
              "1:\n\t"
              "s_bitcmp1_b64 %[sp], %[k]\n\t"
              "s_cbranch_scc1 2f\n\t"
              "s_bitset1_b64 %[vf], %[k]\n\t"
              "v_readlane_b32 %[t], %[len], %[k]\n\t"
              "s_add_u32 %[k], %[k], %[t]\n\t"
              "s_cmp_lt_u32 %[k], %[lim]\n\t"
              "s_cbranch_scc1 1b\n"
              "2:"
              : [k] "+s"(k), [vf] "+s"(visF), [t] "=&s"(hop_t)
              : [sp] "s"(special), [lim] "s"(limit), [len] "v"(len1)
              : "scc");
          cur = k;
          if (cur >= limit) break;
        }
        // the positions of this round that are in the chains so far (bit i: position c0 + i; px - c0 < 64)
        const unsigned long long in_round = final_bits_lo(visF) | Rlo;
        // ---- the lane at `cur` is not final: resolve it now that everything before it is
        const uint32_t px = c0 + cur;  // the position the walk is at
        uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)best1, (int)cur);
        uint32_t td = (uint32_t)__builtin_amdgcn_readlane((int)bd1, (int)cur);
        const uint32_t ts = (uint32_t)__builtin_amdgcn_readlane((int)st1, (int)cur);
        uint32_t maxlen = be - px;
        maxlen = maxlen > kMaxMatch ? kMaxMatch : maxlen;
        if (ts != ST_DONE) {
          // again, now that everything before px is final -- this time with a lane per candidate: entry k of the
          // position's list goes to lane k (evaluating the list once more in the one lane that holds it, entry after
          // entry, cost 160 instructions a time and half of the kernel), and when the chain goes on beyond the list,
          // the list of its last candidate holds the next K candidates (distances from that candidate).  Lengths:
          // the position's own list has them (nibbles); candidates of later trips are compared byte by byte.  A trip
          // beyond the first is two dependent fetches (the list, then the candidates' bytes); the list of the NEXT
          // trip is fetched as soon as this trip's last candidate is known, beside the bytes.
          // A lane that only needs more candidates (its list ran out with the walk still going, none of its entries
          // inside the round) has had its own list evaluated for good in phase 1: it starts with the list of its last
          // candidate, with the best match, the depth left and the last distance that phase 1 ended on.  (Two of three
          // resolutions are of this kind; taking them through the first trip again was a quarter of the kernel.)
          const bool have = lane < (uint32_t)K;
          const bool go_on = ts == ST_MORE;  // (uniform)
          uint32_t e = 0xFFFFu, Ln = 0;
          if (!go_on) {
            uint32_t v = 0, nw = 0;
#pragma unroll
            for (int i = 0; i < K / 2; i++) {
              const uint32_t di = (uint32_t)__builtin_amdgcn_readlane((int)dd[i], (int)cur);
              v = (lane >> 1) == (uint32_t)i ? di : v;
            }
#pragma unroll
            for (int i = 0; i < K / 8; i++) {
              const uint32_t ni = (uint32_t)__builtin_amdgcn_readlane((int)nb[i], (int)cur);
              nw = (lane >> 3) == (uint32_t)i ? ni : nw;
            }
            if (have) e = (v >> (16 * (lane & 1))) & 0xFFFFu;
            Ln = ((nw >> (4 * (lane & 7))) & 15u) + 3;
            Ln = Ln > maxlen ? maxlen : Ln;
          }
          uint32_t dleft = max_chain, cl = px;
          uint32_t my0 = 0, my1 = 0, my2 = 0, my3 = 0;
          bool first_trip = true;
          if (go_on) {  // (tb, td: phase 1's)
            dleft = (uint32_t)__builtin_amdgcn_readlane((int)depth1, (int)cur);
            cl = px - (uint32_t)__builtin_amdgcn_readlane((int)lastd1, (int)cur);
            if (have) e = (uint32_t)dg16[(size_t)cl * K + lane];
            my0 = gld4(V, px + 3), my1 = gld4(V, px + 7), my2 = gld4(V, px + 11), my3 = gld4(V, px + 15);
            first_trip = false;
          } else {
            tb = 0, td = 0;
          }
          uint32_t e_next = 0xFFFFu;
          for (;;) {
            const uint32_t e_last = (uint32_t)__builtin_amdgcn_readlane((int)e, K - 1);
            const bool more = e_last != 0xFFFFu && (e_last >> 15) != 0;
            if (more) {  // (uniform) the next trip's list; harmless if this trip ends the walk
              const uint32_t cl_next = cl - ((e_last & 0x7FFFu) + 1);
              if (have) e_next = (uint32_t)dg16[(size_t)cl_next * K + lane];
            }
            const uint32_t q = cl - ((e & 0x7FFFu) + 1);
            const bool ok = have && e != 0xFFFFu && px - q <= kWindow;  // LD:1365 against px's window
            const unsigned long long notok = ~ballot64(ok);
            const uint32_t first_bad = (uint32_t)__builtin_ctzll(notok);  // (<= K: lanes K.. are never ok)
            const bool in_chain = lane < first_bad;
            const uint32_t qr = q - c0;  // (q < px: below 64 when the candidate lies in this round, huge otherwise)
            const uint32_t wq = qr < 64u ? (uint32_t)(in_round >> (qr & 63u)) : (s_ins[(q >> 5) & 1023u] >> (q & 31));
            const bool ins = in_chain && (wq & 1u);
            const unsigned long long insm = ballot64(ins);
            const bool sel = ins && (uint32_t)__popcll(insm & below) < dleft;  // LD:1362
            uint32_t L = 0;
            if (first_trip) {  // (uniform)
              L = sel ? Ln : 0u;
            } else if (sel) {
              // (the compiler sinks each load into the branch that needs it; forcing all four into flight together was
              // measured: 44.0 ms against 41.4 -- the kernel is bound by what it issues, not by these latencies)
              const uint32_t x0 = gld4(V, q + 3) ^ my0, x1 = gld4(V, q + 7) ^ my1, x2 = gld4(V, q + 11) ^ my2, x3 = gld4(V, q + 15) ^ my3;
              L = 3 + (x0 ? (uint32_t)__builtin_ctz(x0) >> 3
                          : (x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3)
                                : (x2 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : (x3 ? 12u + ((uint32_t)__builtin_ctz(x3) >> 3) : 16u))));
              L = L > maxlen ? maxlen : L;
              L = L > kCandLenCap ? kCandLenCap : L;
            }
            const unsigned long long selm = ballot64(sel);
            const unsigned long long nicem = ballot64(sel && L >= nice);
            const uint32_t first_nice = nicem ? (uint32_t)__builtin_ctzll(nicem) : 64u;
            unsigned long long m = selm & (first_nice >= 63 ? ~0ull : ((2ull << first_nice) - 1ull));
            while (m) {  // (a handful of lanes: nearest first, LD:1385 strict)
              const uint32_t j = (uint32_t)__builtin_ctzll(m);
              m &= m - 1;
              const uint32_t Lj = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)j);
              if (Lj > tb) {
                tb = Lj;
                td = px - (uint32_t)__builtin_amdgcn_readlane((int)q, (int)j);
              }
            }
            dleft -= (uint32_t)__popcll(selm);
            if (nicem || first_bad < (uint32_t)K || dleft == 0 || !more) break;
            cl = (uint32_t)__builtin_amdgcn_readlane((int)q, K - 1);
            e = e_next;
            if (first_trip) {
              my0 = gld4(V, px + 3), my1 = gld4(V, px + 7), my2 = gld4(V, px + 11), my3 = gld4(V, px + 15);
              first_trip = false;
            }
          }
        }
        // a length of 18 is "18 or more": the true length from the bytes, by the whole wave (LD:1371-1383)
        if (tb >= kCandLenCap && maxlen > kCandLenCap) {
          const uint32_t cj = px - td;
          const uint32_t off = 16 + 4 * lane;  // (16 + 4 * 63 > 258: some lane always ends the comparison)
          uint32_t y = 1;
          if (off < maxlen) y = gld4(V, cj + off) ^ gld4(V, px + off);
          const unsigned long long mism = ballot64(y != 0);
          const uint32_t f = (uint32_t)__builtin_ctzll(mism);
          const uint32_t yf = (uint32_t)__builtin_amdgcn_readlane((int)y, (int)f);
          uint32_t Lj = 16 + 4 * f;
          if (Lj < maxlen) Lj += (uint32_t)__builtin_ctz(yf) >> 3;  // (a lane at or beyond maxlen holds y = 1: + 0)
          tb = Lj > maxlen ? maxlen : Lj;
        }
        // ---- the token, and which of its positions enter the chains
        uint32_t len = 1, t;
        bool interior_on = true;
        if (tb > 3 || (tb == 3 && td < 4096)) {
          t = tok_match(tb, td);
          len = tb;
          interior_on = tb <= max_insert;  // LD:1454
        } else {
          t = (uint32_t)__builtin_amdgcn_readlane((int)litv, (int)cur);
        }
        asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(tokv) : "s"(t), "s"(cur) : "m0");
        visR |= 1ull << cur;
        {  // its bits: the first position always, the others when the match is short enough (len <= 6 then)
          const unsigned long long m = interior_on ? (1ull << len) - 1ull : 1ull;
          Rlo |= m << cur;
          Rhi |= cur ? m >> (64u - cur) : 0ull;
        }
        cur += len;
      }
      // ---- the round's tokens, in order, and its bitmap words over [c0, c0 + cur)
      {
        const unsigned long long vis = visF | visR;
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(vis >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vis, 0));
        if (__builtin_amdgcn_inverse_ballot_w64(vis)) tok[ntok + before] = tokv;
        ntok += (uint32_t)__popcll(vis);
        unsigned long long mlo = final_bits_lo(visF) | Rlo, mhi = Rhi;
#pragma unroll
        for (uint32_t j = 1; j < 6; j++)
          if (j < max_insert) mhi |= (visF & Sj[j - 1]) >> (64 - j);
        const uint32_t first = c0, last = c0 + cur - 1, wi = (first >> 5) + lane;
        if (wi <= (last >> 5)) {
          const uint32_t lo = first > wi * 32 ? first - wi * 32 : 0u;
          const uint32_t hi = last < wi * 32 + 31 ? last - wi * 32 : 31u;
          const uint32_t mask = (hi == 31 ? ~0u : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
          const int32_t rel0 = (int32_t)(wi * 32) - (int32_t)c0;  // bit 0 of this word, relative to c0: -31 ..
          uint32_t val;
          if (rel0 < 0)
            val = (uint32_t)(mlo << (uint32_t)(-rel0));
          else if (rel0 < 64)
            val = (uint32_t)(mlo >> (uint32_t)rel0) | (rel0 > 32 ? (uint32_t)(mhi << (uint32_t)(64 - rel0)) : 0u);
          else
            val = rel0 < 128 ? (uint32_t)(mhi >> (uint32_t)(rel0 - 64)) : 0u;
          const uint32_t slot = wi & 1023u;
          s_ins[slot] = (s_ins[slot] & ~mask) | (val & mask);
        }
        lds_order();
      }
      x = c0 + cur;
    }
    if (lane == 0) B.blocks[(size_t)s * B.max_blocks + b].ntok = ntok;
  }
}

__global__ void k_tokens_literal(Batch B) {
  const uint32_t s = blockIdx.x;
  const uint32_t n = stream_len(B, s), D = B.dlen;
  const uint8_t *V = B.vbuf + (size_t)s * (B.nvp + kVPad);
  uint32_t *tok = B.bufA + (size_t)s * B.nvp;
  for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) tok[q] = V[D + q];
  const uint32_t nblk = num_blocks(n);
  for (uint32_t b = threadIdx.x; b < nblk; b += blockDim.x)
    B.blocks[(size_t)s * B.max_blocks + b].ntok = n ? block_end(b, n) - block_start(b) + 1 : 0;
}

}  // namespace ldh
