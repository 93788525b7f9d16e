// k_misc.h -- north-star part (6): chunked-reduction Adler-32 and the 6-bit print codec.
//
//   k_adler32        LibDeflate:Adler32 (LD:401-442)
//   k_print6_encode  LibDeflate:EncodeForPrint (LD:3311-3353)
//   k_print6_decode  LibDeflate:DecodeForPrint (LD:3365-3417)
#pragma once
#include "ldh_common.h"

namespace ldh {

// a = 1 + sum x_j ; b = n + sum (n - j) x_j   (mod 65521), j 0-based: the closed form of
// the per-byte recurrence a=(a+x)%65521, b=(b+a)%65521.  One workgroup per stream.  The
// 16-byte-aligned body of a stream is read 16 bytes per lane, coalesced (4 KiB per trip); for
// the 16 bytes at offset j:  sum (n-j-k) x_k = (n-j) * sum x_k - sum k x_k, with the byte sums
// from v_sad_u8 and the k-weighted sums from v_dot4_u32_u8.  The bytes before and after the
// aligned body (at most 15 each) are taken one per lane.
constexpr int kAdlerThreads = 256;
constexpr uint32_t kAdlerMod = 65521u;
__global__ __launch_bounds__(kAdlerThreads) void k_adler32(const uint8_t *base, const uint64_t *off,
                                                           uint64_t stride, const uint32_t *len_arr,
                                                           uint32_t *adler) {
  // two addressing modes: off != nullptr -> stream s = base[off[s] .. off[s+1]);
  // otherwise stream s = base + s*stride, length len_arr[s].
  __shared__ uint32_t sa[kAdlerThreads / kWave], sb[kAdlerThreads / kWave];
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  const uint8_t *p;
  uint64_t n;
  if (off) {
    p = base + off[s];
    n = off[s + 1] - off[s];
  } else {
    p = base + (uint64_t)s * stride;
    n = len_arr[s];
  }
  uint64_t head = (16 - ((uintptr_t)p & 15)) & 15;
  if (head > n) head = n;
  const uint64_t nvec = (n - head) / 16, tail0 = head + nvec * 16;
  // A = sum x, W = sum ((n - j) mod 65521) * (sum of the bytes at j..), K = sum k x_k inside 16-byte groups
  uint64_t A = 0, W = 0, K = 0;
  if (tid < head) {
    const uint32_t x = p[tid];
    A += x;
    W += (uint64_t)((n - tid) % kAdlerMod) * x;
  }
  if (tail0 + tid < n) {
    const uint32_t x = p[tail0 + tid];
    A += x;
    W += (uint64_t)((n - tail0 - tid) % kAdlerMod) * x;
  }
  if (tid < nvec) {
    const uint4 *v = (const uint4 *)(p + head);
    uint32_t wgt = (uint32_t)((n - head - 16ull * tid) % kAdlerMod);  // weight of the group's first byte
    constexpr uint32_t kStep = (16u * kAdlerThreads) % kAdlerMod;     // 4096
    for (uint64_t g = tid; g < nvec; g += kAdlerThreads) {
      const uint4 x = v[g];
      const uint32_t s0 = __builtin_amdgcn_sad_u8(x.x, 0u, 0u), s1 = __builtin_amdgcn_sad_u8(x.y, 0u, 0u);
      const uint32_t s2 = __builtin_amdgcn_sad_u8(x.z, 0u, 0u), s3 = __builtin_amdgcn_sad_u8(x.w, 0u, 0u);
      uint32_t k = __builtin_amdgcn_udot4(x.x, 0x03020100u, 0u, false);
      k = __builtin_amdgcn_udot4(x.y, 0x07060504u, k, false);
      k = __builtin_amdgcn_udot4(x.z, 0x0b0a0908u, k, false);
      k = __builtin_amdgcn_udot4(x.w, 0x0f0e0d0cu, k, false);
      const uint32_t sum = s0 + s1 + s2 + s3;
      A += sum;
      W += (uint64_t)(wgt * sum);  // < 65521 * 4080 < 2^32
      K += k;
      wgt = wgt >= kStep ? wgt - kStep : wgt + kAdlerMod - kStep;
    }
  }
  // b's share of this lane: W - K (mod 65521)
  uint32_t a32 = (uint32_t)(A % kAdlerMod);
  uint32_t b32 = (uint32_t)((W % kAdlerMod + kAdlerMod - K % kAdlerMod) % kAdlerMod);
  for (int o = 32; o > 0; o >>= 1) {  // 64 values below 2^16 each
    a32 += __shfl_down(a32, o);
    b32 += __shfl_down(b32, o);
  }
  if ((tid & 63) == 0) {
    sa[tid >> 6] = a32;
    sb[tid >> 6] = b32;
  }
  __syncthreads();
  if (tid == 0) {
    uint64_t ta = 1, tb = n % kAdlerMod;
    for (int w = 0; w < kAdlerThreads / kWave; w++) {
      ta += sa[w];
      tb += sb[w];
    }
    adler[s] = (uint32_t)(((tb % kAdlerMod) << 16) | (ta % kAdlerMod));
  }
}

__device__ inline uint8_t p6_char(uint32_t v) {  // LD:3165-3230
  return (uint8_t)(v < 26 ? 'a' + v : (v < 52 ? 'A' + (v - 26) : (v < 62 ? '0' + (v - 52) : (v == 62 ? '(' : ')'))));
}
__device__ inline int p6_value(uint32_t c) {  // LD:3232-3297
  if (c >= 'a' && c <= 'z') return (int)(c - 'a');
  if (c >= 'A' && c <= 'Z') return (int)(c - 'A') + 26;
  if (c >= '0' && c <= '9') return (int)(c - '0') + 52;
  if (c == '(') return 62;
  if (c == ')') return 63;
  return -1;
}

constexpr int kP6Threads = 256;
// One workgroup per stream; a thread turns 3 bytes into 4 characters.
__global__ __launch_bounds__(kP6Threads) void k_print6_encode(const uint8_t *in, const uint64_t *off,
                                                              uint8_t *out, uint64_t out_stride,
                                                              uint32_t *out_len) {
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  const uint8_t *p = in + off[s];
  const uint64_t n = off[s + 1] - off[s];
  uint8_t *o = out + (uint64_t)s * out_stride;
  const uint64_t groups = n / 3;
  // twelve bytes -> sixteen characters per thread and trip: aligned dword loads shifted into place
  // (the input starts at any byte), one 16-byte store (out slots are 16-byte aligned)
  const uint64_t quads = groups / 4;
  {
    const uintptr_t src = (uintptr_t)p;
    const uint32_t sh = (uint32_t)(src & 3);
    const uint32_t *w32 = (const uint32_t *)(src - sh);
    uint4 *o128 = (uint4 *)o;
    for (uint64_t g4 = tid; g4 < quads; g4 += kP6Threads) {
      const uint32_t *q = w32 + 3 * g4;
      const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = sh ? q[3] : 0u;  // (sh == 0: q[3] may lie past the input)
      const uint32_t a = __builtin_amdgcn_alignbyte(w1, w0, sh), bq = __builtin_amdgcn_alignbyte(w2, w1, sh),
                     cq = __builtin_amdgcn_alignbyte(w3, w2, sh);
      const uint32_t c24[4] = {a & 0xFFFFFFu, (a >> 24) | ((bq & 0xFFFFu) << 8), (bq >> 16) | ((cq & 0xFFu) << 16), cq >> 8};
      uint32_t out4[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t c = c24[k];
        out4[k] = (uint32_t)p6_char(c & 63) | ((uint32_t)p6_char((c >> 6) & 63) << 8) |
                  ((uint32_t)p6_char((c >> 12) & 63) << 16) | ((uint32_t)p6_char((c >> 18) & 63) << 24);
      }
      o128[g4] = make_uint4(out4[0], out4[1], out4[2], out4[3]);
    }
  }
  for (uint64_t g = quads * 4 + tid; g < groups; g += kP6Threads) {
    const uint32_t c = (uint32_t)p[3 * g] | ((uint32_t)p[3 * g + 1] << 8) | ((uint32_t)p[3 * g + 2] << 16);
    const uint32_t w = (uint32_t)p6_char(c & 63) | ((uint32_t)p6_char((c >> 6) & 63) << 8) |
                       ((uint32_t)p6_char((c >> 12) & 63) << 16) | ((uint32_t)p6_char((c >> 18) & 63) << 24);
    *(uint32_t *)(o + 4 * g) = w;  // out slots are 16-byte aligned
  }
  if (tid == 0) {  // LD:3336-3350 tail: 1 byte -> 2 chars, 2 bytes -> 3 chars
    const uint64_t rem = n - groups * 3;
    uint32_t cache = 0;
    for (uint64_t k = 0; k < rem; k++) cache |= (uint32_t)p[groups * 3 + k] << (8 * k);
    uint64_t w = groups * 4;
    int bits = (int)rem * 8;
    while (bits > 0) {
      o[w++] = p6_char(cache & 63);
      cache >>= 6;
      bits -= 6;
    }
    out_len[s] = (uint32_t)w;
  }
}

__global__ __launch_bounds__(kP6Threads) void k_print6_decode(const uint8_t *in, const uint64_t *off,
                                                              uint8_t *out, uint64_t out_stride,
                                                              uint32_t *out_len, uint8_t *ok) {
  __shared__ unsigned long long s_lo, s_hi;
  __shared__ uint32_t s_bad;
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  const uint8_t *p = in + off[s];
  const uint64_t n = off[s + 1] - off[s];
  uint8_t *o = out + (uint64_t)s * out_stride;
  if (tid == 0) {
    s_lo = n;
    s_hi = 0;
    s_bad = 0;
  }
  __syncthreads();
  // strip "^[%c ]+" and "[%c ]+$" (LD:3370-3371): first / one-past-last kept byte
  // (scanned from both ends, 256 bytes at a time, until a kept byte shows: normally one trip each,
  // instead of a pass over the whole input)
  for (uint64_t c0 = 0; c0 < n; c0 += kP6Threads) {
    const uint64_t i = c0 + tid;
    bool keep = false;
    if (i < n) {
      const uint32_t c = p[i];
      keep = !(c < 32 || c == 127 || c == ' ');
      if (keep) atomicMin(&s_lo, (unsigned long long)i);
    }
    if (__syncthreads_or(keep)) break;
  }
  for (uint64_t c0 = 0; c0 < n; c0 += kP6Threads) {
    const uint64_t k = c0 + tid;
    bool keep = false;
    if (k < n) {
      const uint32_t c = p[n - 1 - k];
      keep = !(c < 32 || c == 127 || c == ' ');
      if (keep) atomicMax(&s_hi, (unsigned long long)(n - k));
    }
    if (__syncthreads_or(keep)) break;
  }
  __syncthreads();
  const uint64_t b = s_lo < s_hi ? s_lo : 0, e = s_lo < s_hi ? s_hi : 0;
  const uint64_t m = e - b;
  if (m == 1) {  // LD:3374
    if (tid == 0) {
      ok[s] = 0;
      out_len[s] = 0;
    }
    return;
  }
  const uint64_t groups = m / 4;
  uint32_t bad = 0;
  // sixteen characters -> twelve bytes per thread and trip: aligned dword loads (shifted into place with
  // v_alignbyte: the kept part starts at any byte) and three dword stores (out slots are 16-byte aligned)
  const uint64_t quads = groups / 4;
  {
    const uintptr_t src = (uintptr_t)(p + b);
    const uint32_t sh = (uint32_t)(src & 3);
    const uint32_t *w32 = (const uint32_t *)(src - sh);
    uint32_t *o32 = (uint32_t *)o;
    for (uint64_t g4 = tid; g4 < quads; g4 += kP6Threads) {
      const uint32_t *q = w32 + 4 * g4;
      uint32_t w[5];
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = q[k];
      w[4] = sh ? q[4] : 0u;  // (with sh == 0 the fifth dword is not part of the group and may lie past the input)
      uint32_t c24[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t ch = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
        const int x1 = p6_value(ch & 255), x2 = p6_value((ch >> 8) & 255), x3 = p6_value((ch >> 16) & 255), x4 = p6_value(ch >> 24);
        if ((x1 | x2 | x3 | x4) < 0) bad = 1;
        c24[k] = ((uint32_t)x1 | ((uint32_t)x2 << 6) | ((uint32_t)x3 << 12) | ((uint32_t)x4 << 18)) & 0xFFFFFFu;
      }
      o32[3 * g4] = c24[0] | (c24[1] << 24);
      o32[3 * g4 + 1] = (c24[1] >> 8) | (c24[2] << 16);
      o32[3 * g4 + 2] = (c24[2] >> 16) | (c24[3] << 8);
    }
  }
  for (uint64_t g = quads * 4 + tid; g < groups; g += kP6Threads) {
    const uint8_t *q = p + b + 4 * g;
    const int x1 = p6_value(q[0]), x2 = p6_value(q[1]), x3 = p6_value(q[2]), x4 = p6_value(q[3]);
    if ((x1 | x2 | x3 | x4) < 0) {
      bad = 1;
    } else {
      const uint32_t c = (uint32_t)x1 | ((uint32_t)x2 << 6) | ((uint32_t)x3 << 12) | ((uint32_t)x4 << 18);
      o[3 * g] = (uint8_t)c;
      o[3 * g + 1] = (uint8_t)(c >> 8);
      o[3 * g + 2] = (uint8_t)(c >> 16);
    }
  }
  if (bad) atomicOr(&s_bad, 1u);
  __syncthreads();
  if (tid == 0) {  // LD:3397-3414 tail
    uint32_t isbad = s_bad;
    const uint64_t rem = m - groups * 4;
    uint32_t cache = 0;
    int bits = 0;
    uint64_t w = groups * 3;
    for (uint64_t k = 0; k < rem; k++) {
      const int x = p6_value(p[b + groups * 4 + k]);
      if (x < 0) {
        isbad = 1;
        break;
      }
      cache |= (uint32_t)x << bits;
      bits += 6;
    }
    if (!isbad) {
      while (bits >= 8) {
        o[w++] = (uint8_t)cache;
        cache >>= 8;
        bits -= 8;
      }
    }
    ok[s] = isbad ? 0 : 1;
    out_len[s] = isbad ? 0u : (uint32_t)w;
  }
}


// ------------------------------------------------------------------ CreateCodec codecs
// LibDeflate:CreateCodec (LD:2921-3056) and, through it, Encode/DecodeForWoWAddonChannel / ChatChannel
// (LD:3058-3158).  The Lua closure's gsub tables become byte tables in device memory:
struct CodecTables {
  uint8_t enc_len[256];  // 1: byte (possibly mapped) -> enc0;  2: escape byte enc0 + suffix enc1
  uint8_t enc0[256], enc1[256];
  uint8_t reserved[256];   // decode_fail_pattern (LD:3040)
  uint8_t map_back[256];   // first decode gsub (identity where nothing is mapped)
  uint8_t pass_escape[256];
  uint32_t npass;
  uint32_t pad_;
  uint16_t pass_dec[1][256];  // npass tables: suffix -> original byte, 0xFFFF = not in the set
};

constexpr int kCodecThreads = 256;
constexpr uint32_t kCodecPer = 16;  // consecutive bytes per thread per tile

__device__ inline uint32_t codec_block_scan(uint32_t val, uint32_t tid, uint32_t *s_w, uint32_t *tile_total) {
  const uint32_t lane = tid & 63, wave = tid >> 6;
  uint32_t inc = val;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= (uint32_t)off) inc += t;
  }
  __syncthreads();  // s_w of the tile before has been read
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < kCodecThreads / kWave; w++) {
    if (w < wave) base += s_w[w];
    tot += s_w[w];
  }
  *tile_total = tot;
  return base + inc - val;
}

// codec:Encode (LD:3030-3037): one gsub that replaces every byte to translate by one or two bytes.
// One workgroup per stream; expansion is compacted with a scan per 4096-byte tile.
__global__ __launch_bounds__(kCodecThreads) void k_codec_encode(const uint8_t *in, const uint64_t *in_off,
                                                               const CodecTables *T, uint8_t *out,
                                                               uint64_t out_stride, uint32_t *out_len) {
  __shared__ uint32_t s_w[kCodecThreads / kWave];
  __shared__ uint32_t s_enc[256];  // enc0 | enc1 << 8 | enc_len << 16 (per byte they were three gathers from global memory)
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  const uint8_t *p = in + in_off[s];
  const uint64_t n = in_off[s + 1] - in_off[s];
  uint8_t *o = out + (uint64_t)s * out_stride;
  for (uint32_t i = tid; i < 256; i += kCodecThreads)
    s_enc[i] = (uint32_t)T->enc0[i] | ((uint32_t)T->enc1[i] << 8) | ((uint32_t)T->enc_len[i] << 16);
  __syncthreads();
  uint64_t obase = 0;
  for (uint64_t t0 = 0; t0 < n; t0 += kCodecThreads * kCodecPer) {
    const uint64_t j0 = t0 + (uint64_t)tid * kCodecPer;
    uint32_t e[kCodecPer];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < kCodecPer; k++) {
      e[k] = j0 + k < n ? s_enc[p[j0 + k]] : 0u;
      cnt += e[k] >> 16;
    }
    uint32_t tot;
    uint64_t w = obase + codec_block_scan(cnt, tid, s_w, &tot);
#pragma unroll
    for (uint32_t k = 0; k < kCodecPer; k++)
      if (e[k] >> 16) {
        o[w++] = (uint8_t)e[k];
        if ((e[k] >> 16) == 2) o[w++] = (uint8_t)(e[k] >> 8);
      }
    obase += tot;
  }
  if (tid == 0) out_len[s] = (uint32_t)obase;
}

// codec:Decode (LD:3042-3053): nil if a reserved byte is present; then one gsub per pass over the
// whole string.  A launch is one pass: pass 0 reads the input (through the single-byte translate of
// the optional first gsub, which is per byte and can be applied on the fly), later passes read the
// previous pass's slots.  "escape + suffix" matches cannot overlap (the escape byte is never a
// suffix), so whether a byte starts a pair, or is the second byte of one, is decided locally.
__global__ __launch_bounds__(kCodecThreads) void k_codec_decode_pass(const uint8_t *in, const uint64_t *in_off,
                                                                    const uint8_t *prev, const uint32_t *prev_len,
                                                                    uint64_t stride, const CodecTables *T,
                                                                    uint32_t pass, uint8_t *out, uint32_t *out_len,
                                                                    uint8_t *ok) {
  __shared__ uint32_t s_w[kCodecThreads / kWave];
  __shared__ uint32_t s_bad;
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  const bool first = pass == 0;
  const uint8_t *p = first ? in + in_off[s] : prev + (uint64_t)s * stride;
  const uint64_t n = first ? in_off[s + 1] - in_off[s] : prev_len[s];
  uint8_t *o = out + (uint64_t)s * stride;
  // this pass's tables in LDS: dec (escape suffix -> byte, 0xFFFF none) | map_back << 16 | reserved << 24
  __shared__ uint32_t s_tab[256];
  if (!first && !ok[s]) {
    if (tid == 0) out_len[s] = 0;
    return;
  }
  if (tid == 0) s_bad = 0;
  for (uint32_t i = tid; i < 256; i += kCodecThreads)
    s_tab[i] = (uint32_t)T->pass_dec[pass][i] | ((uint32_t)T->map_back[i] << 16) | ((uint32_t)T->reserved[i] << 24);
  __syncthreads();
  const uint8_t E = T->pass_escape[pass];
  uint32_t bad = 0;  // (first pass) a reserved byte anywhere in the input: nil (LD:3048), found while reading
  uint64_t obase = 0;
  for (uint64_t t0 = 0; t0 < n; t0 += kCodecThreads * kCodecPer) {
    const uint64_t j0 = t0 + (uint64_t)tid * kCodecPer;
    // bytes j0-1 .. j0+kCodecPer (one before, one after), through the single-byte translate in pass 0
    uint8_t b[kCodecPer + 2];
#pragma unroll
    for (uint32_t k = 0; k < kCodecPer + 2; k++) {
      const uint64_t j = j0 + k;  // index + 1
      uint8_t x = 0;
      if (j >= 1 && j - 1 < n) {
        x = p[j - 1];
        if (first) {
          const uint32_t t = s_tab[x];
          bad |= t >> 24;
          x = (uint8_t)(t >> 16);
        }
      }
      b[k] = x;
    }
    uint8_t val[kCodecPer];
    uint32_t keepm = 0, cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < kCodecPer; k++) {
      const uint64_t j = j0 + k;
      const uint8_t x = b[k + 1];
      const uint32_t dn = s_tab[b[k + 2]] & 0xFFFFu, dx = s_tab[x] & 0xFFFFu;
      const bool starts = j + 1 < n && x == E && dn != 0xFFFF;
      const bool second = j >= 1 && b[k] == E && dx != 0xFFFF;
      val[k] = starts ? (uint8_t)dn : x;
      if (j < n && !second) {
        keepm |= 1u << k;
        cnt++;
      }
    }
    uint32_t tot;
    uint64_t w = obase + codec_block_scan(cnt, tid, s_w, &tot);
#pragma unroll
    for (uint32_t k = 0; k < kCodecPer; k++)
      if ((keepm >> k) & 1) o[w++] = val[k];
    obase += tot;
  }
  if (first) {
    if (bad) s_bad = 1;
    __syncthreads();
    if (tid == 0) {
      ok[s] = s_bad ? 0 : 1;
      out_len[s] = s_bad ? 0u : (uint32_t)obase;
    }
    return;
  }
  if (tid == 0) out_len[s] = (uint32_t)obase;
}

}  // namespace ldh
