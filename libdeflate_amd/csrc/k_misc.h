// k_misc.h -- north-star part (6): chunked-reduction Adler-32 and the 6-bit print codec.
//
//   k_adler32        LibDeflate:Adler32 (LD:401-442)
//   k_print6_encode  LibDeflate:EncodeForPrint (LD:3311-3353)
//   k_print6_decode  LibDeflate:DecodeForPrint (LD:3365-3417)
#pragma once
#include "ldh_common.h"

namespace ldh {

// a = 1 + sum x_j ; b = n + sum (n - j) x_j   (mod 65521), j 0-based: the closed form of
// the per-byte recurrence a=(a+x)%65521, b=(b+a)%65521.  One workgroup per stream; every
// thread reduces a contiguous chunk, then the workgroup sums.
constexpr int kAdlerThreads = 256;
__global__ __launch_bounds__(kAdlerThreads) void k_adler32(const uint8_t *base, const uint64_t *off,
                                                           uint64_t stride, const uint32_t *len_arr,
                                                           uint32_t *adler) {
  // two addressing modes: off != nullptr -> stream s = base[off[s] .. off[s+1]);
  // otherwise stream s = base + s*stride, length len_arr[s].
  __shared__ uint64_t sa[kAdlerThreads / kWave], sb[kAdlerThreads / kWave];
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
  const uint64_t per = (n + kAdlerThreads - 1) / kAdlerThreads;
  const uint64_t j0 = per * tid, j1 = (j0 + per < n) ? j0 + per : n;
  uint64_t A = 0, Bw = 0;
  if (j0 < n) {
    // weight of byte j is (n - j); reduce the weight mod 65521 so products stay < 2^24
    uint64_t wgt = (n - j0) % 65521u;
    uint32_t guard = 0;
    for (uint64_t j = j0; j < j1; j++) {
      const uint32_t x = p[j];
      A += x;
      Bw += wgt * x;
      wgt = wgt ? wgt - 1 : 65520u;
      if (++guard == (1u << 20)) {
        A %= 65521u;
        Bw %= 65521u;
        guard = 0;
      }
    }
    A %= 65521u;
    Bw %= 65521u;
  }
  for (int o = 32; o > 0; o >>= 1) {
    A += __shfl_down(A, o);
    Bw += __shfl_down(Bw, o);
  }
  if ((tid & 63) == 0) {
    sa[tid >> 6] = A;
    sb[tid >> 6] = Bw;
  }
  __syncthreads();
  if (tid == 0) {
    uint64_t ta = 1, tb = n % 65521u;
    for (int w = 0; w < kAdlerThreads / kWave; w++) {
      ta += sa[w];
      tb += sb[w];
    }
    adler[s] = (uint32_t)(((tb % 65521u) << 16) | (ta % 65521u));
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
  for (uint64_t g = tid; g < groups; g += kP6Threads) {
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
  unsigned long long lo = n, hi = 0;
  for (uint64_t i = tid; i < n; i += kP6Threads) {
    const uint32_t c = p[i];
    if (!(c < 32 || c == 127 || c == ' ')) {
      if (i < lo) lo = i;
      if (i + 1 > hi) hi = i + 1;
    }
  }
  atomicMin(&s_lo, lo);
  atomicMax(&s_hi, hi);
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
  for (uint64_t g = tid; g < groups; g += kP6Threads) {
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

}  // namespace ldh
