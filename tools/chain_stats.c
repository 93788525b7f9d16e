// chain_stats.c -- analysis only (not product, not oracle): how many candidates does the lazy
// search (LD:1349-1398, depth = max_chain, Q3) look at per position, and how many of them
// share 4 / 5 / 6 bytes with the position?  Decides whether k-gram links pay in k_match_lazy.
//   gcc -O2 -o /tmp/chain_stats tools/chain_stats.c && /tmp/chain_stats file stream_len max_chain nice
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  FILE *f = fopen(argv[1], "rb");
  const uint32_t n = (uint32_t)atoi(argv[2]), max_chain = (uint32_t)atoi(argv[3]), nice = (uint32_t)atoi(argv[4]);
  uint8_t *b = calloc(n + 300, 1);
  if (!f || fread(b, 1, n, f) != n) return 3;
  int32_t *head = malloc(sizeof(int32_t) << 24), *prev = malloc(sizeof(int32_t) * n);
  memset(head, 0xFF, sizeof(int32_t) << 24);
  uint64_t searched = 0, steps = 0, k4 = 0, k5 = 0, k6 = 0, improve = 0, pass2 = 0, visited = 0;
  uint64_t steps_v = 0, k4_v = 0;
  uint32_t *L = calloc(n + 2, 4), *Dd = calloc(n + 2, 4);
  for (uint32_t x = 0; x + 2 < n; x++) {  // (single block: fine for statistics)
    const uint32_t h = (b[x] << 16) | (b[x + 1] << 8) | b[x + 2];
    int32_t c = head[h];
    prev[x] = c;
    head[h] = (int32_t)x;
    uint32_t best = 0, depth = max_chain, maxlen = n - x < 258 ? n - x : 258, bd = 0, s = 0, s4 = 0;
    searched++;
    while (c >= 0 && depth > 0 && x - (uint32_t)c <= 32768) {
      uint32_t l = 3;
      while (l < maxlen && b[c + l] == b[x + l]) l++;
      steps++, s++;
      if (l >= 4) k4++, s4++;
      if (l >= 5) k5++;
      if (l >= 6) k6++;
      if (best >= 1 && b[c + best - 1] == b[x + best - 1] && b[c + best] == b[x + best]) pass2++;
      if (l > best) best = l, bd = x - (uint32_t)c, improve++;
      if (best >= nice) break;
      c = prev[c];
      depth--;
    }
    L[x] = best, Dd[x] = bd;
    (void)s, (void)s4;
  }
  // the lazy walk (successor function form): which positions does the parse look at?
  const uint32_t max_lazy = max_chain == 32 ? 16 : (max_chain == 16 ? 4 : (max_chain <= 256 ? (max_chain == 128 ? 16 : 32) : 128));
  uint8_t *need = calloc(n + 2, 1);
  for (uint32_t x = 0; x < n;) {
    need[x] = 1;
    const uint32_t pl = L[x], pd = Dd[x];
    uint32_t look = 0;
    if (pl < max_lazy && x + 1 < n) look = L[x + 1], need[x + 1] = 1;
    const int usable = pl > 3 || (pl == 3 && pd < 4096);
    x += (usable && look <= pl) ? pl : 1;
  }
  for (uint32_t x = 0; x < n; x++) visited += need[x];
  (void)steps_v, (void)k4_v;
  printf("positions %u searched %llu steps/pos %.2f  4-gram %.2f  5-gram %.2f  6-gram %.2f  improve %.2f  reject-pass %.2f  needed-by-parse %.1f%%\n",
         n, (unsigned long long)searched, (double)steps / searched, (double)k4 / searched, (double)k5 / searched,
         (double)k6 / searched, (double)improve / searched, (double)pass2 / searched, 100.0 * visited / n);
  return 0;
}
