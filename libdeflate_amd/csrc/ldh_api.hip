// ldh_api.hip -- host side of libdeflate_hip.so: the C ABI of include/libdeflate_hip.h
// over the gfx950 kernels.  No CPU fallback exists: every entry point needs a HIP device.
#include "../../include/libdeflate_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <pthread.h>
#include <sched.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "k_huffman.h"
#include "k_inflate.h"
#include "k_lz77.h"
#include "k_misc.h"
#include "ldh_common.h"

using namespace ldh;

#ifndef LDH_NOALLOC_MAX_STREAMS
#define LDH_NOALLOC_MAX_STREAMS 65536  // LDH_NOALLOC_MAX_STREAMS of the header: 2 x 4 bytes of static device memory each
#endif

struct ldh_dict {
  std::vector<uint8_t> bytes;  // host copy: every device a context runs on gets its own upload
  uint32_t adler = 0;
  std::mutex mu;
  std::vector<std::pair<int, uint8_t *>> dev;  // (device, bytes + 64 zeros)
};

struct ldh_codec {
  std::vector<uint8_t> img;  // host image of ldh::CodecTables: header + npass suffix tables
  uint32_t npass = 0;
  std::mutex mu;
  std::vector<std::pair<int, ldh::CodecTables *>> dev;
};

// ---- the no-allocation entry point's scratch (ldh_decompress_batch_dev_noalloc) --------------------
// File-scope device memory, one copy per GPU, held until the process ends: the zlib trailer values k_inflate reads
// and the Adler-32 sums k_adler32 computes for kNoAllocMaxStreams streams.  In host code this name is only a
// stand-in; each context keeps the address hipGetSymbolAddress gave for its device (Ctx::noalloc).
constexpr size_t kNoAllocMaxStreams = LDH_NOALLOC_MAX_STREAMS;
static __device__ __align__(256) uint32_t scratch[2 * kNoAllocMaxStreams];

namespace {

thread_local std::string g_err;
std::atomic<uint64_t> g_allocs{0};  // hipMalloc / hipHostMalloc calls the library has made (ldh_debug_alloc_count)

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
};
// one piece in flight of the host-pointer pipelines (see below)
struct Slot {
  PinBuf pin_in, pin_off, pin_out, pin_small;
  DevBuf d_in, d_off, d_out, d_len, d_a8, d_a32a, d_a32b, d_scr;  // d_scr: a piece's zlib trailer values and Adler-32 sums
  hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
  size_t w = 0;  // row width of the result copy in flight
};
struct ProfEntry {
  std::string name;
  double ms = 0;
  uint64_t launches = 0;
};
struct Pending {
  int idx;
  hipEvent_t a, b;
};

// Everything the library owns on one GPU: streams, grow-only workspace, staging, profiler.  ldh_init(device) makes
// one; ldh_init_devices() makes a pool of them (one per listed device; a device may be listed twice) over which the
// host-pointer batch calls split their streams.  Calls on one context are serialised by its mutex.
struct Ctx {
  int device = -1;
  std::mutex mu;
  hipStream_t stream = nullptr, s_in = nullptr, s_out = nullptr, s_k2 = nullptr;  // s_k2: every second inflate piece's kernels
  // workspace (grow-only)
  DevBuf vbuf, prevd, bufA, bufB, bufC, blocks, adler, misc32a, misc32b;
  // staging for the host-pointer entry points
  DevBuf h_in, h_off, h_out, h_len, h_aux8, h_aux32a, h_aux32b, codec_tmp;
  Slot slot[5];  // (compress pipelines two pieces; inflate keeps up to three in flight, see decompress_batch_on)
  uint32_t *flags = nullptr;    // [0] streams longer than max_in_len since the last ldh_sync (k_plan)
  uint32_t *flags_host = nullptr;  // pinned word ldh_sync reads flags[0] into, on cx.stream
  uint32_t *noalloc = nullptr;  // this device's copy of `scratch`
  hipEvent_t ev_noalloc = nullptr;  // end of this context's last ldh_decompress_batch_dev_noalloc (see g_scratch_last)
  bool attr_done = false;
  std::vector<int> near_cpus;   // the CPUs of this GPU's NUMA node (threads the library starts for this context run there)
  // last compress batch (for ldh_debug_fetch)
  Batch last{};
  bool have_last = false;
  // profiler
  bool prof = false;
  std::vector<ProfEntry> prof_entries;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t t0 = nullptr, t1 = nullptr;
};

std::mutex g_mu;                 // guards the context tables below
Ctx *g_default = nullptr;        // the context of ldh_init(device): what every call without a pool uses
std::vector<Ctx *> g_pool;       // ldh_init_devices(): contexts the host-pointer batch calls are split over
thread_local Ctx *t_cur = nullptr;  // ldh_use_context(): this thread's choice among the pool
// `scratch` is one array per GPU, but a GPU may carry several contexts (ldh_init_devices({0, 0})), each with a stream of
// its own: per device, the context whose no-allocation call used the array last and the event that marks its end.  The
// next such call from another context makes its stream wait for that event first (no allocation, no host wait).
std::mutex g_scratch_mu;
std::vector<std::pair<int, Ctx *>> g_scratch_last;  // (device, context)

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(LDH_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

Ctx *current() { return t_cur ? t_cur : g_default; }
#define CTX_ENTER_ON(cx, ptr)                                                \
  Ctx *cxp_ = (ptr);                                                         \
  if (!cxp_) return fail(LDH_E_NOT_INIT, "ldh_init() has not been called"); \
  Ctx &cx = *cxp_;                                                           \
  std::lock_guard<std::mutex> lk(cx.mu);                                     \
  HIPCHK(hipSetDevice(cx.device))
// every entry point: CTX_ENTER(cx) -> Ctx &cx, locked, its device current in this thread
#define CTX_ENTER(cx) CTX_ENTER_ON(cx, current())

int ensure(DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    hipDeviceSynchronize();  // (the host-pointer entry points keep copies in flight on other streams)
    hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 4096;
  g_allocs++;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    want = bytes + 4096;
    e = hipMalloc(&b.p, want);
  }
  if (e != hipSuccess) return fail(LDH_E_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
  b.cap = want;
  return 0;
}

hipEvent_t get_event(Ctx &cx) {
  if (!cx.event_pool.empty()) {
    hipEvent_t e = cx.event_pool.back();
    cx.event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  g_allocs++;
  hipEventCreate(&e);
  return e;
}
int prof_index(Ctx &cx, const char *name) {
  for (size_t i = 0; i < cx.prof_entries.size(); i++)
    if (cx.prof_entries[i].name == name) return (int)i;
  cx.prof_entries.push_back({name, 0, 0});
  return (int)cx.prof_entries.size() - 1;
}
void prof_drain(Ctx &cx) {
  if (cx.pending.empty()) return;
  hipStreamSynchronize(cx.stream);
  for (auto &p : cx.pending) {
    float ms = 0;
    hipEventElapsedTime(&ms, p.a, p.b);
    cx.prof_entries[p.idx].ms += ms;
    cx.prof_entries[p.idx].launches++;
    cx.event_pool.push_back(p.a);
    cx.event_pool.push_back(p.b);
  }
  cx.pending.clear();
}

struct ProfScope {
  Ctx &cx;
  int idx = -1;
  hipEvent_t a = nullptr;
  ProfScope(Ctx &c, const char *name) : cx(c) {
    if (cx.prof) {
      idx = prof_index(cx, name);
      a = get_event(cx);
      hipEventRecord(a, cx.stream);
    }
  }
  ~ProfScope() {
    if (idx >= 0) {
      hipEvent_t b = get_event(cx);
      hipEventRecord(b, cx.stream);
      cx.pending.push_back({idx, a, b});
      if (cx.pending.size() > 4096) prof_drain(cx);
    }
  }
};

#define LAUNCH(name, kern, grid, block, shmem, ...)                                   \
  do {                                                                                \
    ProfScope ps_(cx, name);                                                          \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, cx.stream, __VA_ARGS__); \
  } while (0)

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

size_t workspace_budget() {
  const char *e = getenv("LDH_WORKSPACE_MB");
  size_t mb = e ? (size_t)strtoull(e, nullptr, 10) : 32768;
  if (mb < 64) mb = 64;
  return mb << 20;
}

// the device copy of a dictionary / codec on this context's GPU (uploaded on first use there)
const uint8_t *dict_on(Ctx &cx, const ldh_dict *dc) {
  ldh_dict *d = const_cast<ldh_dict *>(dc);
  std::lock_guard<std::mutex> lk(d->mu);
  for (auto &e : d->dev)
    if (e.first == cx.device) return e.second;
  uint8_t *p = nullptr;
  const size_t n = d->bytes.size();
  g_allocs++;
  if (hipMalloc((void **)&p, n + 64) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, n + 64) != hipSuccess || hipMemcpy(p, d->bytes.data(), n, hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(p);
    return nullptr;
  }
  d->dev.push_back({cx.device, p});
  return p;
}
const ldh::CodecTables *codec_on(Ctx &cx, const ldh_codec *cc) {
  ldh_codec *c = const_cast<ldh_codec *>(cc);
  std::lock_guard<std::mutex> lk(c->mu);
  for (auto &e : c->dev)
    if (e.first == cx.device) return e.second;
  ldh::CodecTables *p = nullptr;
  g_allocs++;
  if (hipMalloc((void **)&p, c->img.size()) != hipSuccess) return nullptr;
  if (hipMemcpy(p, c->img.data(), c->img.size(), hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(p);
    return nullptr;
  }
  c->dev.push_back({cx.device, p});
  return p;
}

// ---- caller-pinned host memory, thread placement, host-side trace --------------------------------------------
// Ranges the caller has pinned through ldh_host_register / ldh_host_alloc.  A host-pointer batch call whose inputs (or
// outputs) lie inside one of them copies straight between the caller's memory and the device: no gather into the
// library's pinned staging, no scatter out of it.
struct HostRange {
  uintptr_t lo, hi;
  bool owned;  // hipHostMalloc'ed by ldh_host_alloc (else hipHostRegister'ed)
};
std::mutex g_host_mu;
std::vector<HostRange> g_host_ranges;
bool host_pinned(const void *p, size_t bytes) {
  if (!bytes) return true;
  const uintptr_t a = (uintptr_t)p;
  std::lock_guard<std::mutex> lk(g_host_mu);
  for (const HostRange &r : g_host_ranges)
    if (a >= r.lo && a + bytes <= r.hi) return true;
  return false;
}

// The CPUs next to a GPU ("pin threads and buffers per NUMA node"): the PCI device's numa_node in sysfs and
// that node's cpulist.  Best effort: no sysfs entry, node -1 or LDH_NO_AFFINITY=1 leave the threads where they are.
std::vector<int> cpus_near_device(int device) {
  std::vector<int> cpus;
  if (getenv("LDH_NO_AFFINITY")) return cpus;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return cpus;
  for (char *c = bus; *c; c++) *c = (char)tolower(*c);
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE *f = fopen(path, "r");
  int node = -1;
  if (f) {
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
  }
  if (node < 0) return cpus;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  if (!f) return cpus;
  char line[4096] = {0};
  if (fgets(line, sizeof(line), f)) {
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
      int a = 0, b = 0;
      const int k = sscanf(tok, "%d-%d", &a, &b);
      if (k == 1) b = a;
      if (k >= 1)
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) cpus.push_back(c);
    }
  }
  fclose(f);
  // (never widen what the process was given: cgroups / taskset of the caller stay in force)
  cpu_set_t cur;
  CPU_ZERO(&cur);
  if (sched_getaffinity(0, sizeof(cur), &cur) == 0) {
    std::vector<int> both;
    for (int c : cpus)
      if (CPU_ISSET(c, &cur)) both.push_back(c);
    cpus.swap(both);
  }
  return cpus;
}
void bind_this_thread(const std::vector<int> &cpus) {
  if (cpus.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus) CPU_SET(c, &set);
  pthread_setaffinity_np(pthread_self(), sizeof(set), &set);  // (failure: the thread stays where it is)
}
thread_local const std::vector<int> *t_near = nullptr;  // the calling context's CPU list, for the copy threads it starts

// LDH_HOST_TRACE=1: host-clock milliseconds per stage of the host-pointer pipelines, printed to stderr per call
struct HostTrace {
  bool on = getenv("LDH_HOST_TRACE") != nullptr;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now(), start = last;
  void lap(int k) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    t[k] += std::chrono::duration<double, std::milli>(now - last).count();
    last = now;
  }
  void report(const char *what, size_t pieces, bool din, bool dout) {
    if (!on) return;
    const double tot = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count();
    fprintf(stderr,
            "[ldh host trace] %s: %zu piece(s), input %s, output %s: total %.2f ms = gather/enqueue-in %.2f + enqueue kernels %.2f + "
            "wait kernels %.2f + wait copy-out %.2f + scatter %.2f + other %.2f\n",
            what, pieces, din ? "direct" : "staged", dout ? "direct" : "staged", tot, t[0], t[1], t[2], t[3], t[4],
            tot - t[0] - t[1] - t[2] - t[3] - t[4]);
  }
};

// ---- compress ---------------------------------------------------------------
int compress_dev_locked(Ctx &cx, const uint8_t *d_in, const uint64_t *d_in_off, size_t count, size_t max_in_len,
                        int fmt, int level, int strategy, const ldh_dict *dict, uint8_t *d_out,
                        size_t out_stride, uint32_t *d_out_len, uint8_t *d_pad_bits) {
  if (count == 0) return 0;
  if (level < 0 || level > 9) return fail(LDH_E_ARG, "level must be 0..9 for *_dev (got %d)", level);
  if (strategy < 0 || strategy > 3) return fail(LDH_E_ARG, "bad strategy %d", strategy);
  if (fmt != 0 && fmt != 1) return fail(LDH_E_ARG, "bad fmt %d", fmt);
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  if (max_in_len > 0xFFFF0000ull) return fail(LDH_E_ARG, "streams of 4 GiB or more are not supported");
  if (out_stride < ldh_compress_bound(max_in_len))
    return fail(LDH_E_CAPACITY, "out_stride %zu < ldh_compress_bound(%zu) = %zu", out_stride, max_in_len,
                ldh_compress_bound(max_in_len));
  const uint32_t dlen = dict ? (uint32_t)dict->bytes.size() : 0;
  const bool need_lz = level != 0 && strategy != LDH_STRATEGY_HUFFMAN_ONLY;
  // the dictionary only matters to the match search; keep V = input otherwise
  const uint32_t vd = need_lz ? dlen : 0;
  const uint32_t nvp = (uint32_t)round_up((size_t)vd + max_in_len + 16, 16);
  const uint32_t max_blocks = num_blocks((uint32_t)max_in_len);

  const bool greedy = need_lz && !level_cfg(level).lazy;  // levels 1-3
  // level 1 without a dictionary: candidate lists for every position (k_cand_lists), then one wavefront per stream
  // that takes 64 positions a round (k_lz_greedy2); levels 2-3 and batches with a dictionary: k_lz_greedy on the sorted array
  static const bool greedy_v1 = getenv("LDH_GREEDY_V1") != nullptr;
  // (level 2, whose walk wants 8 inserted candidates, measured slower this way than on the sorted array: LDH_GREEDY2_MAX=2
  // turns it on for that level too, with lists of 16)
  static const int greedy2_max = getenv("LDH_GREEDY2_MAX") ? atoi(getenv("LDH_GREEDY2_MAX")) : 1;
  const bool greedy2 = greedy && level <= greedy2_max && level <= 2 && vd == 0 && !greedy_v1;
  const int candK = level == 1 ? 8 : 16;
  if (greedy2) {  // k_lz_greedy2<K> has its level's parameters built in
    const LevelCfg lc = level_cfg(level);
    if (lc.nice != (candK == 8 ? 8 : 18) || lc.max_chain != (candK == 8 ? 4 : 8) || lc.max_lazy != (candK == 8 ? 4 : 5))
      return fail(LDH_E_ARG, "k_lz_greedy2: level %d does not match the constants of K = %d", level, candK);
  }
  const bool want_links = need_lz && (!greedy || greedy2);  // prevd[]: the lazy levels' search and the candidate lists
  // streams beyond the narrow sort's: k_seg_links (segments of 32768 positions sorted inside LDS, any length)
  const bool seg_links = want_links && ((size_t)nvp + kVPad + 15) / 16 * 16 > kSortNarrowMaxLds;
  // (a workgroup takes a run of consecutive segments: whole streams when there are streams enough to fill the
  // chip -- up to 64 segments, 2 MiB -- else runs of kSegRun, whose first segment is sorted twice)
  const uint32_t seg_max = (uint32_t)(((size_t)vd + max_in_len + kSegG - 1) / kSegG);
  const uint32_t seg_run_len = (count >= 1024 && seg_max <= 64) ? std::max(seg_max, 1u) : kSegRun;
  const uint32_t seg_runs = (seg_max + seg_run_len - 1) / seg_run_len;
  // the rest (greedy levels, short streams, V alone for level 0 / huffman_only): k_sort_links, in LDS when V fits there
  const bool sort_in_lds = (size_t)nvp + kVPad <= kSortLdsMaxV;
  const size_t cand_bytes = greedy2 ? cand_stream_bytes(nvp, candK) : 0;
  const size_t per_stream = (size_t)(nvp + kVPad) + 2ull * nvp + ((greedy && !greedy2) ? 24ull : 8ull) * nvp + cand_bytes +
                            (size_t)max_blocks * sizeof(BlockInfo) + 4;
  size_t chunk = std::max<size_t>(1, workspace_budget() / per_stream);
  if (chunk > count) chunk = count;

  if (int rc = ensure(cx.vbuf, chunk * (size_t)(nvp + kVPad) + 64)) return rc;
  if (int rc = ensure(cx.prevd, chunk * 2ull * nvp + 64)) return rc;
  if (int rc = ensure(cx.bufA, chunk * 4ull * nvp + 64)) return rc;
  if (int rc = ensure(cx.bufB, chunk * 4ull * nvp + 64)) return rc;
  if (greedy)
    if (int rc = ensure(cx.bufC, chunk * (greedy2 ? cand_bytes : 16ull * nvp) + 64)) return rc;
  if (int rc = ensure(cx.blocks, chunk * (size_t)max_blocks * sizeof(BlockInfo))) return rc;
  if (int rc = ensure(cx.adler, chunk * 4 + 64)) return rc;

  const uint8_t *d_dict = nullptr;
  if (dict && !(d_dict = dict_on(cx, dict))) return fail(LDH_E_NOMEM, "dictionary upload to device %d failed", cx.device);
  if (!cx.attr_done) {
    HIPCHK(hipFuncSetAttribute((const void *)k_sort_links<true, kSortThreads>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(kSortLdsMaxV)));
    HIPCHK(hipFuncSetAttribute((const void *)k_seg_links, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSegLdsBytes));
    for (const void *f : {(const void *)k_match_lazy<false, kRefillMinShallow, kMatchThreads>, (const void *)k_match_lazy<true, kRefillMinShallow, kMatchThreads>,
                          (const void *)k_match_lazy<false, kRefillMinDeep, kMatchThreads>,
                          (const void *)k_match_lazy<true, kRefillMinDeep, kMatchThreads>})
      HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMatchLdsBytes));
    for (const void *f : {(const void *)k_cand_lists<8, kCandThreads>, (const void *)k_cand_lists<16, kCandThreads>})
      HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCandLdsBytes));
    cx.attr_done = true;
  }

  for (size_t s0 = 0; s0 < count; s0 += chunk) {
    const uint32_t c = (uint32_t)std::min(chunk, count - s0);
    Batch B{};
    B.in = d_in;
    B.in_off = d_in_off + s0;
    B.count = c;
    B.dict = vd ? d_dict : nullptr;
    B.dlen = vd;
    B.dict_adler = dict ? dict->adler : 0;
    B.max_len = (uint32_t)max_in_len;
    B.nvp = nvp;
    B.max_blocks = max_blocks;
    B.level = level;
    B.strategy = strategy;
    B.fmt = fmt;
    B.vbuf = (uint8_t *)cx.vbuf.p;
    B.prevd = (uint16_t *)cx.prevd.p;
    B.bufA = (uint32_t *)cx.bufA.p;
    B.bufB = (uint32_t *)cx.bufB.p;
    B.bufC = greedy ? (uint32_t *)cx.bufC.p : nullptr;
    B.want_rank = (greedy && !greedy2) ? 1u : 0u;
    B.sort_lds = sort_in_lds ? ((nvp + kVPad + 15) & ~15u) : 0u;
    B.need_lz = need_lz ? 1u : 0u;
    B.blocks = (BlockInfo *)cx.blocks.p;
    B.adler = (uint32_t *)cx.adler.p;
    B.out = d_out + s0 * out_stride;
    B.out_stride = out_stride;
    B.out_len = d_out_len + s0;
    B.pad_bits = d_pad_bits + s0;
    B.too_long = cx.flags;
    // k_plan / k_bitpack read the zlib DICTID even when V carries no dictionary
    Batch Bz = B;
    Bz.dict = d_dict;

    if (getenv("LDH_POISON")) {  // test hook: no kernel may depend on what a buffer held before
      HIPCHK(hipMemsetAsync(cx.vbuf.p, 0xA5, cx.vbuf.cap, cx.stream));
      HIPCHK(hipMemsetAsync(cx.prevd.p, 0xA5, cx.prevd.cap, cx.stream));
      HIPCHK(hipMemsetAsync(cx.bufA.p, 0xA5, cx.bufA.cap, cx.stream));
      HIPCHK(hipMemsetAsync(cx.bufB.p, 0xA5, cx.bufB.cap, cx.stream));
      if (greedy) HIPCHK(hipMemsetAsync(cx.bufC.p, 0xA5, cx.bufC.cap, cx.stream));
      HIPCHK(hipMemsetAsync(cx.blocks.p, 0xA5, cx.blocks.cap, cx.stream));
      HIPCHK(hipMemsetAsync(B.out, 0xA5, (size_t)c * out_stride, cx.stream));
    }
    // V (and, for the match search, the trigram chains)
    if (seg_links)
      LAUNCH("k_seg_links", k_seg_links, c * seg_runs, kSegThreads, kSegLdsBytes, B, seg_runs, seg_run_len);
    else if (B.sort_lds && B.sort_lds <= kSortNarrowMaxLds)
      LAUNCH("k_sort_links", (k_sort_links<true, kSortThreadsNarrow>), c, kSortThreadsNarrow, B.sort_lds, B);
    else if (B.sort_lds)
      LAUNCH("k_sort_links", (k_sort_links<true, kSortThreads>), c, kSortThreads, B.sort_lds, B);
    else
      LAUNCH("k_sort_links", (k_sort_links<false, kSortThreads>), c, kSortThreads, 0, B);
    if (level != 0) {
      if (strategy == LDH_STRATEGY_HUFFMAN_ONLY) {
        LAUNCH("k_tokens_literal", k_tokens_literal, c, 256, 0, B);
      } else if (level_cfg(level).lazy) {
        const uint32_t tiles = (uint32_t)((max_in_len + kTile - 1) / kTile);
        const bool deep = level_cfg(level).max_chain >= 256;
        if (tiles && (size_t)vd + max_in_len <= kMatchNarrowMaxV) {
          // short streams: one tile is the whole stream; small workgroups, LDS sized to the longest stream
          const uint32_t cap = (uint32_t)round_up((size_t)vd + max_in_len + 288 + 16, 16);
          const uint32_t lds = cap + 2 * (uint32_t)(vd + max_in_len + 8);
          if (vd && !deep) LAUNCH("k_match_lazy", (k_match_lazy<true, kRefillMinShallow, kMatchThreadsNarrow>), c * tiles, kMatchThreadsNarrow, lds, B, tiles, cap, 1u);
          if (!vd && !deep) LAUNCH("k_match_lazy", (k_match_lazy<false, kRefillMinShallow, kMatchThreadsNarrow>), c * tiles, kMatchThreadsNarrow, lds, B, tiles, cap, 1u);
          if (vd && deep) LAUNCH("k_match_lazy", (k_match_lazy<true, kRefillMinDeep, kMatchThreadsNarrow>), c * tiles, kMatchThreadsNarrow, lds, B, tiles, cap, 1u);
          if (!vd && deep) LAUNCH("k_match_lazy", (k_match_lazy<false, kRefillMinDeep, kMatchThreadsNarrow>), c * tiles, kMatchThreadsNarrow, lds, B, tiles, cap, 1u);
        } else if (tiles) {
          const uint32_t cap = kLdsByteCap;
          // a workgroup takes a run of consecutive tiles (it keeps the window in LDS from one to the next): whole
          // streams when the batch has workgroups enough without cutting them, else four tiles
          const uint32_t tpw = count >= 1024 ? tiles : (tiles < 4 ? tiles : 4u);
          const uint32_t mruns = (tiles + tpw - 1) / tpw;
          if (vd && !deep) LAUNCH("k_match_lazy", (k_match_lazy<true, kRefillMinShallow, kMatchThreads>), c * mruns, kMatchThreads, kMatchLdsBytes, B, tiles, cap, tpw);
          if (!vd && !deep) LAUNCH("k_match_lazy", (k_match_lazy<false, kRefillMinShallow, kMatchThreads>), c * mruns, kMatchThreads, kMatchLdsBytes, B, tiles, cap, tpw);
          if (vd && deep) LAUNCH("k_match_lazy", (k_match_lazy<true, kRefillMinDeep, kMatchThreads>), c * mruns, kMatchThreads, kMatchLdsBytes, B, tiles, cap, tpw);
          if (!vd && deep) LAUNCH("k_match_lazy", (k_match_lazy<false, kRefillMinDeep, kMatchThreads>), c * mruns, kMatchThreads, kMatchLdsBytes, B, tiles, cap, tpw);
        }
#ifndef LDH_PARSE_NARROW_MAX
#define LDH_PARSE_NARROW_MAX (kParseThreadsNarrow * kParseChunk)
#endif
        if (max_in_len <= LDH_PARSE_NARROW_MAX)
          LAUNCH("k_parse_lazy", k_parse_lazy<kParseThreadsNarrow>, c * max_blocks, kParseThreadsNarrow, 0, B);
        else
          LAUNCH("k_parse_lazy", k_parse_lazy<kParseThreads>, c * max_blocks, kParseThreads, 0, B);
      } else if (greedy2) {
        const uint32_t ctiles = (uint32_t)((max_in_len + kCandTile - 1) / kCandTile);
        if (ctiles) {
          if (max_in_len <= kMatchNarrowMaxV) {  // short streams: one tile is the whole stream, LDS sized to the longest
            const uint32_t cap = (uint32_t)round_up(max_in_len + 288 + 16, 16);
            const uint32_t lds = cap + 2 * (uint32_t)(max_in_len + 8);
            if (candK == 8) LAUNCH("k_cand_lists", (k_cand_lists<8, kCandThreadsNarrow>), c * ctiles, kCandThreadsNarrow, lds, B, ctiles, cap);
            else LAUNCH("k_cand_lists", (k_cand_lists<16, kCandThreadsNarrow>), c * ctiles, kCandThreadsNarrow, lds, B, ctiles, cap);
          } else {
            if (candK == 8) LAUNCH("k_cand_lists", (k_cand_lists<8, kCandThreads>), c * ctiles, kCandThreads, kCandLdsBytes, B, ctiles, kCandByteCap);
            else LAUNCH("k_cand_lists", (k_cand_lists<16, kCandThreads>), c * ctiles, kCandThreads, kCandLdsBytes, B, ctiles, kCandByteCap);
          }
        }
        if (candK == 8) LAUNCH("k_lz_greedy", k_lz_greedy2<8>, c, kWave, 0, B);
        else LAUNCH("k_lz_greedy", k_lz_greedy2<16>, c, kWave, 0, B);
      } else {
        LAUNCH("k_lz_greedy", k_lz_greedy, c, kWave, 0, B);
      }
      LAUNCH("k_histogram", k_histogram, c * max_blocks, 256, 0, B);
      LAUNCH("k_huffman", k_huffman, c * max_blocks, kWave, 0, B);
    }
    if (fmt == 1)
      LAUNCH("k_adler32", k_adler32, c, kAdlerThreads, 0, d_in, B.in_off, (uint64_t)0, (const uint32_t *)nullptr, B.adler);
    if (max_blocks > 16)  // streams beyond 1 MiB: a wavefront per stream
      LAUNCH("k_plan", k_plan_wave, c, kWave, 0, Bz);
    else
      LAUNCH("k_plan", k_plan, (c + 63) / 64, 64, 0, Bz);
    LAUNCH("k_bitpack", k_bitpack, c * max_blocks, kPackThreads, 0, Bz);
    HIPCHK(hipGetLastError());
    cx.last = B;
    cx.have_last = true;
  }
  return 0;
}

// ---- decompress -------------------------------------------------------------
int decompress_dev_locked(Ctx &cx, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                          size_t in_stride, size_t count, int fmt,
                          const ldh_dict *dict, uint8_t *d_out, size_t out_stride, uint32_t *d_out_len,
                          uint32_t *d_bytes_left, int32_t *d_status, uint32_t *scratch2 = nullptr,
                          uint32_t *actual2 = nullptr, int force_waves = 0) {
  if (count == 0) return 0;
  if (fmt != 0 && fmt != 1) return fail(LDH_E_ARG, "bad fmt %d", fmt);
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  // zlib trailer values read by k_inflate / Adler-32 sums of the results: library workspace, or (the
  // no-allocation entry point) the two halves of the caller-independent static scratch
  // (actual2: the caller's own second array -- the host path's pieces, which run side by side, each bring theirs)
  uint32_t *expect = scratch2, *actual = actual2 ? actual2 : (scratch2 ? scratch2 + kNoAllocMaxStreams : nullptr);
  if (!scratch2) {
    if (int rc = ensure(cx.misc32a, count * 4 + 64)) return rc;
    if (int rc = ensure(cx.misc32b, count * 4 + 64)) return rc;
    expect = (uint32_t *)cx.misc32a.p;
    actual = (uint32_t *)cx.misc32b.p;
  }
  const uint8_t *d_dict = nullptr;
  if (dict && !(d_dict = dict_on(cx, dict))) return fail(LDH_E_NOMEM, "dictionary upload to device %d failed", cx.device);
  InflateArgs A{};
  A.in = d_in;
  A.in_off = d_in_off;
  A.in_len = d_in_len;
  A.in_stride = in_stride;
  A.count = (uint32_t)count;
  A.fmt = fmt;
  A.dict = d_dict;
  A.dlen = dict ? (uint32_t)dict->bytes.size() : 0;
  A.dict_adler = dict ? dict->adler : 0;
  A.out = d_out;
  A.out_stride = out_stride;
  A.out_len = d_out_len;
  A.bytes_left = d_bytes_left;
  A.status = d_status;
  A.expect_adler = expect;
  // One wavefront per stream when the batch has streams enough to keep the chip busy that way (k_inflate_alone: the block
  // body inlined; k_inflate, with the body as a call, is kept for comparison); fewer streams get a workgroup of four or eight
  // wavefronts each, a round speculating on 256 / 512 bit offsets (decode_block_wide).  Measured on 128 KiB streams
  // (tools/inflate_rate.py, ms for 1 / 256 / 512 / 1024 / 1280 / 2048 streams): one wavefront 3.1 / 3.3 / 3.4 / 3.7 / 4.0 /
  // 4.6, four 2.3 / 2.5 / 2.8 / 3.2 / 5.3 / 6.5, eight 1.6 / 1.8 / 2.3 / 4.4 / 5.6 / 8.8: the wide forms run four
  // wavefronts a SIMD, so 512 x 8 and 1024 x 4 wavefronts are what the chip holds at once, and one stream more starts a
  // second shift.  LDH_INFLATE_WAVES=1/2/4/8 forces a form.
  static const int body_call_min = getenv("LDH_INFLATE_CALL_MIN") ? atoi(getenv("LDH_INFLATE_CALL_MIN")) : (1 << 30);
  static const int wide4_max = getenv("LDH_INFLATE_WIDE4_MAX") ? atoi(getenv("LDH_INFLATE_WIDE4_MAX")) : 1025;
  static const int wide8_max = getenv("LDH_INFLATE_WIDE8_MAX") ? atoi(getenv("LDH_INFLATE_WIDE8_MAX")) : 513;
  const char *force = getenv("LDH_INFLATE_WAVES");
  const int waves = force ? atoi(force) : (force_waves ? force_waves : (count < (size_t)wide8_max ? 8 : (count < (size_t)wide4_max ? 4 : 1)));
  if (waves == 8)
    LAUNCH("k_inflate", k_inflate_wide8, (uint32_t)count, 8 * kWave, 0, A);
  else if (waves == 2)
    LAUNCH("k_inflate", k_inflate_wide2, (uint32_t)count, 2 * kWave, 0, A);
  else if (waves == 4)
    LAUNCH("k_inflate", k_inflate_wide4, (uint32_t)count, 4 * kWave, 0, A);
  else if (count >= (size_t)body_call_min)
    LAUNCH("k_inflate", k_inflate, (uint32_t)count, kWave, 0, A);
  else
    LAUNCH("k_inflate", k_inflate_alone, (uint32_t)count, kWave, 0, A);
  if (fmt == 1) {
    LAUNCH("k_adler32", k_adler32, (uint32_t)count, kAdlerThreads, 0, (const uint8_t *)d_out,
           (const uint64_t *)nullptr, (uint64_t)out_stride, (const uint32_t *)d_out_len, actual);
    LAUNCH("k_zlib_check", k_zlib_check, (uint32_t)((count + 255) / 256), 256, 0, (uint32_t)count,
           (const uint32_t *)expect, (const uint32_t *)actual, d_status);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// host staging helpers ---------------------------------------------------------
int upload_inputs(Ctx &cx, const uint8_t *in_base, const uint64_t *in_off, size_t count) {
  const uint64_t total = in_off[count];
  if (int rc = ensure(cx.h_in, total + 64)) return rc;
  if (int rc = ensure(cx.h_off, (count + 1) * 8)) return rc;
  // offsets are re-based to start at 0 so that the device copy of the data is aligned
  std::vector<uint64_t> off(count + 1);
  for (size_t i = 0; i <= count; i++) off[i] = in_off[i] - in_off[0];
  if (total - in_off[0])
    HIPCHK(hipMemcpyAsync(cx.h_in.p, in_base + in_off[0], total - in_off[0], hipMemcpyHostToDevice, cx.stream));
  HIPCHK(hipMemcpyAsync(cx.h_off.p, off.data(), (count + 1) * 8, hipMemcpyHostToDevice, cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));  // `off` goes out of scope
  return 0;
}

// ---- host-pointer batches: pinned staging, copies overlapped with kernels -----
// A batch is cut into pieces of about piece_bytes() (64 MiB).  Piece k is gathered into pinned memory by
// a few host threads, copied in on one stream, processed on cx.stream, and its results are
// copied out (only the bytes produced: a 2D copy as wide as the piece's longest result) on a
// third stream and scattered to the caller's slots by the host threads -- while the kernels
// of piece k+1 run.  Compression alternates two slots; decompression, whose results are the
// large side, rotates three, so that the copy-out of piece k-1 also overlaps the scatter of
// piece k-2.
// (LDH_PIECE_KB: test hook, so that small batches are cut into many pieces too)
size_t piece_bytes() {
  const char *e = getenv("LDH_PIECE_KB");
  const size_t kb = e ? (size_t)strtoull(e, nullptr, 10) : 65536;
  return std::max<size_t>(kb, 1) << 10;
}
// A stream takes milliseconds to inflate however few there are, so decompression wants many streams per piece: 128 MiB
// of results (1024 streams of 128 KiB: k_inflate_wide's range; 256 MiB and one wavefront per stream before round 3)
// -- for batches up to a few thousand streams.  A large batch is better served by pieces of 256 MiB (2048 streams, one
// wavefront each: 4.6 ms against two wide launches of 3.2): there are pieces enough to hide the copies anyway.
size_t inflate_piece_bytes(size_t total_out) {
  static const size_t forced = getenv("LDH_INFLATE_PIECE_MULT") ? (size_t)atoi(getenv("LDH_INFLATE_PIECE_MULT")) : 0;
  static const size_t big_from = getenv("LDH_INFLATE_BIG_FROM_MB") ? (size_t)atoi(getenv("LDH_INFLATE_BIG_FROM_MB")) : 768;
  const size_t mult = forced ? forced : (total_out >= (big_from << 20) ? 4 : 2);
  return mult * piece_bytes();
}
int ensure_pinned(PinBuf &b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    hipDeviceSynchronize();
    hipHostFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = bytes + bytes / 8 + 4096;
  g_allocs++;
  const hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
  if (e != hipSuccess) return fail(LDH_E_NOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
  b.cap = want;
  return 0;
}

int pipeline_init(Ctx &cx) {
  if (cx.s_in) return 0;
  HIPCHK(hipStreamCreateWithFlags(&cx.s_in, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&cx.s_out, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&cx.s_k2, hipStreamNonBlocking));
  for (Slot &sl : cx.slot) {
    HIPCHK(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
  }
  return 0;
}

void pipeline_free(Ctx &cx) {
  for (Slot &sl : cx.slot) {
    for (PinBuf *b : {&sl.pin_in, &sl.pin_off, &sl.pin_out, &sl.pin_small}) {
      if (b->p) hipHostFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
    for (DevBuf *b : {&sl.d_in, &sl.d_off, &sl.d_out, &sl.d_len, &sl.d_a8, &sl.d_a32a, &sl.d_a32b, &sl.d_scr}) {
      if (b->p) hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
    if (sl.ev_in) hipEventDestroy(sl.ev_in);
    if (sl.ev_done) hipEventDestroy(sl.ev_done);
    if (sl.ev_out) hipEventDestroy(sl.ev_out);
    sl.ev_in = sl.ev_done = sl.ev_out = nullptr;
  }
  if (cx.s_in) hipStreamDestroy(cx.s_in);
  if (cx.s_out) hipStreamDestroy(cx.s_out);
  if (cx.s_k2) hipStreamDestroy(cx.s_k2);
  cx.s_in = cx.s_out = cx.s_k2 = nullptr;
}

// an entry point that returns early (an error) must not leave copies or kernels in flight on
// the slots the next call will reuse
// the context's kernel stream replaced for a scope (LAUNCH and the profiler's events follow cx.stream)
struct StreamSwap {
  Ctx &cx;
  hipStream_t keep;
  StreamSwap(Ctx &c, hipStream_t s) : cx(c), keep(c.stream) { cx.stream = s; }
  void restore() { cx.stream = keep; }
  ~StreamSwap() { restore(); }
};
struct DrainGuard {
  bool clean = false;
  ~DrainGuard() {
    if (!clean) hipDeviceSynchronize();
  }
};

unsigned copy_threads() {
  static unsigned n = 0;
  if (!n) {
    const char *e = getenv("LDH_COPY_THREADS");
    n = e ? (unsigned)strtoul(e, nullptr, 10) : 8;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && n > hw) n = hw;
    if (n < 1) n = 1;
  }
  return n;
}
// f(lo, hi) over [0, n) on the copy threads (inline when the job is small)
template <class F>
void par_ranges(size_t n, size_t bytes, F f) {
  const unsigned T = bytes < (2u << 20) ? 1u : copy_threads();
  if (T <= 1 || n < 2) {
    f((size_t)0, n);
    return;
  }
  const size_t per = (n + T - 1) / T;
  std::vector<std::thread> th;
  const std::vector<int> *near = t_near;
  for (size_t lo = per; lo < n; lo += per)
    th.emplace_back([=] {
      if (near) bind_this_thread(*near);
      f(lo, std::min(n, lo + per));
    });
  f((size_t)0, std::min(n, per));
  for (auto &t : th) t.join();
}
void par_memcpy(void *dst, const void *src, size_t bytes) {
  constexpr size_t kBlk = 1u << 20;
  par_ranges((bytes + kBlk - 1) / kBlk, bytes, [=](size_t lo, size_t hi) {
    const size_t a = lo * kBlk, b = std::min(bytes, hi * kBlk);
    memcpy((uint8_t *)dst + a, (const uint8_t *)src + a, b - a);
  });
}

}  // namespace

// =============================================================================
extern "C" {

namespace {
void ctx_destroy(Ctx *c);
// g_mu held.  Returns the context, or nullptr with *rc = the error code (g_err holds the message); whatever a failed
// creation had made so far is released again.
Ctx *ctx_create(int device, int *rc) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    *rc = fail(LDH_E_HIP, "no HIP device: %s", hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= n) {
    *rc = fail(LDH_E_ARG, "device %d out of range (count %d)", device, n);
    return nullptr;
  }
  Ctx *c = new Ctx;
  c->device = device;
  void *sym = nullptr;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&c->t0)) != hipSuccess || (e = hipEventCreate(&c->t1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_noalloc, hipEventDisableTiming)) != hipSuccess ||
      (e = hipMalloc((void **)&c->flags, 256)) != hipSuccess || (e = hipMemset(c->flags, 0, 256)) != hipSuccess ||
      (e = hipHostMalloc((void **)&c->flags_host, 64, hipHostMallocDefault)) != hipSuccess ||
      // this GPU's copy of the no-allocation scratch: its address here, never the array's name, is what kernels get
      (e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch))) != hipSuccess) {
    *rc = fail(LDH_E_HIP, "context on device %d: %s", device, hipGetErrorString(e));
    ctx_destroy(c);
    delete c;
    return nullptr;
  }
  c->noalloc = (uint32_t *)sym;
  c->near_cpus = cpus_near_device(device);
  {
    std::lock_guard<std::mutex> sl(g_scratch_mu);
    g_scratch_last.reserve(64);
    bool have = false;
    for (auto &en : g_scratch_last) have = have || en.first == device;
    if (!have) g_scratch_last.push_back({device, nullptr});
  }
  *rc = 0;
  return c;
}
void ctx_destroy(Ctx *c) {
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  hipSetDevice(c->device);
  if (c->stream) hipDeviceSynchronize();
  Ctx &cx = *c;
  pipeline_free(cx);
  for (DevBuf *b : {&cx.vbuf, &cx.prevd, &cx.bufA, &cx.bufB, &cx.bufC, &cx.blocks, &cx.adler, &cx.misc32a, &cx.misc32b, &cx.h_in,
                    &cx.h_off, &cx.h_out, &cx.h_len, &cx.h_aux8, &cx.h_aux32a, &cx.h_aux32b, &cx.codec_tmp}) {
    if (b->p) hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
  }
  for (auto &p : cx.pending) {
    hipEventDestroy(p.a);
    hipEventDestroy(p.b);
  }
  for (hipEvent_t e : cx.event_pool) hipEventDestroy(e);
  cx.pending.clear();
  cx.event_pool.clear();
  {
    std::lock_guard<std::mutex> sl(g_scratch_mu);
    for (auto &e : g_scratch_last)
      if (e.second == c) e.second = nullptr;
  }
  if (cx.flags) hipFree(cx.flags);
  if (cx.flags_host) hipHostFree(cx.flags_host);
  if (cx.ev_noalloc) hipEventDestroy(cx.ev_noalloc);
  if (cx.t0) hipEventDestroy(cx.t0);
  if (cx.t1) hipEventDestroy(cx.t1);
  if (cx.stream) hipStreamDestroy(cx.stream);
  cx.flags = cx.flags_host = nullptr;
  cx.ev_noalloc = cx.t0 = cx.t1 = nullptr;
  cx.stream = nullptr;
}
}  // namespace

int ldh_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_default && g_default->device == device) return 0;
  for (Ctx *c : g_pool)  // a pool exists (ldh_init_devices): its first context on this device becomes the default
    if (c->device == device) {
      g_default = c;
      return 0;
    }
  if (g_default)
    return fail(LDH_E_ARG, "already initialised on device %d (ldh_init_devices() drives several)", g_default->device);
  int rc = 0;
  Ctx *c = ctx_create(device, &rc);
  if (!c) return rc;
  g_default = c;
  return 0;
}

int ldh_init_devices(const int *devices, int n) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_pool.empty()) return fail(LDH_E_ARG, "ldh_init_devices() was already called (%zu contexts)", g_pool.size());
  std::vector<int> list;
  if (devices && n > 0) {
    list.assign(devices, devices + n);
  } else {
    const int cnt = ldh_device_count();
    if (cnt <= 0) return fail(LDH_E_HIP, "no HIP device");
    for (int d = 0; d < cnt; d++) list.push_back(d);
  }
  std::vector<Ctx *> made;
  bool default_used = false;
  for (int d : list) {
    Ctx *c = nullptr;
    if (g_default && g_default->device == d && !default_used) {
      c = g_default;
      default_used = true;
    } else {
      int rc = 0;
      c = ctx_create(d, &rc);
      if (!c) {
        const std::string msg = g_err;  // (ctx_destroy below makes HIP calls of its own)
        for (Ctx *m : made)
          if (m != g_default) {
            ctx_destroy(m);
            delete m;
          }
        g_err = msg;
        return rc;
      }
    }
    made.push_back(c);
  }
  g_pool = made;
  if (!g_default) g_default = g_pool[0];
  return 0;
}

int ldh_pool_size(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_pool.size();
}

int ldh_use_context(int index) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (index < 0) {
    t_cur = nullptr;
    return 0;
  }
  if ((size_t)index >= g_pool.size()) return fail(LDH_E_ARG, "context %d out of range (pool of %zu)", index, g_pool.size());
  t_cur = g_pool[index];
  return 0;
}

void ldh_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  bool default_in_pool = false;
  for (Ctx *c : g_pool) {
    default_in_pool = default_in_pool || c == g_default;
    ctx_destroy(c);
    delete c;
  }
  g_pool.clear();
  if (g_default && !default_in_pool) {
    ctx_destroy(g_default);
    delete g_default;
  }
  g_default = nullptr;
  t_cur = nullptr;  // (other threads' choices dangle: shut down only when no other thread uses the library)
}

int ldh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *ldh_last_error(void) { return g_err.c_str(); }
void *ldh_stream(void) {
  Ctx *c = current();
  return c ? (void *)c->stream : nullptr;
}
int ldh_sync(void) {
  CTX_ENTER(cx);
  // streams a *_dev compress call found longer than its max_in_len (k_plan counts them; their out_len is 0).  The
  // counter is read and cleared ON the library stream (the stream is non-blocking: a null-stream memset would not be
  // ordered against a compress call enqueued right after this one returns, and could wipe out its report)
  HIPCHK(hipMemcpyAsync(cx.flags_host, cx.flags, 4, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));
  const uint32_t too_long = *(volatile uint32_t *)cx.flags_host;
  if (too_long) {
    HIPCHK(hipMemsetAsync(cx.flags, 0, 4, cx.stream));
    return fail(LDH_E_TOO_LONG, "%u stream(s) longer than max_in_len were not compressed (out_len 0)", too_long);
  }
  return 0;
}

size_t ldh_compress_bound(size_t n) {
  // forced fixed/dynamic strategies can exceed the stored size (<= ~9.2 bits per literal
  // plus a dynamic header per block); rounded so it can be used as out_stride.
  size_t nblocks = n <= kFirstBlock ? 1 : 1 + (n - kFirstBlock + kNextBlock - 1) / kNextBlock;
  return round_up(n + n / 4 + 512 * nblocks + 64, 16);
}

// Contiguous ranges of a batch balanced by bytes: part p takes streams [first[p], first[p+1]).
// Pure host arithmetic (tested without a GPU).
int ldh_partition_by_bytes(const uint64_t *in_off, size_t count, int parts, size_t *first) {
  if (!in_off || !first || parts <= 0) return fail(LDH_E_ARG, "ldh_partition_by_bytes: bad arguments");
  const uint64_t base = in_off[0], total = in_off[count] - base;
  first[0] = 0;
  size_t s = 0;
  for (int p = 1; p < parts; p++) {
    // the boundary of part p is the first stream that starts at or beyond p/parts of the bytes (streams are
    // whole; an empty batch or zero-length streams go to the earliest part that can take them)
    const uint64_t target = (uint64_t)((__uint128_t)total * (uint64_t)p / (uint64_t)parts);
    while (s < count && in_off[s] - base < target) s++;
    // spread streams evenly when bytes do not separate them (all empty, or fewer bytes than parts)
    first[p] = s;
  }
  first[parts] = count;
  if (total == 0)
    for (int p = 1; p < parts; p++) first[p] = count * (size_t)p / (size_t)parts;
  return 0;
}

ldh_dict *ldh_dict_create(const uint8_t *bytes, size_t n) {
  Ctx *cxp = current();
  if (!cxp) {
    fail(LDH_E_NOT_INIT, "ldh_init() has not been called");
    return nullptr;
  }
  Ctx &cx = *cxp;
  std::lock_guard<std::mutex> lk(cx.mu);
  if (!bytes || n == 0 || n > 32768) {  // LD:513-521
    fail(LDH_E_ARG, "dictionary length must be 1..32768 (got %zu)", n);
    return nullptr;
  }
  if (hipSetDevice(cx.device) != hipSuccess) {
    fail(LDH_E_HIP, "hipSetDevice(%d)", cx.device);
    return nullptr;
  }
  ldh_dict *d = new ldh_dict;
  d->bytes.assign(bytes, bytes + n);
  const uint8_t *d_bytes = dict_on(cx, d);
  if (!d_bytes) {
    delete d;
    fail(LDH_E_NOMEM, "hipMalloc dict");
    return nullptr;
  }
  // Adler-32 of the dictionary through the device kernel (LD:522)
  uint64_t off[2] = {0, n};
  uint64_t *d_off = nullptr;
  uint32_t *d_ad = nullptr;
  hipMalloc((void **)&d_off, 16);
  hipMalloc((void **)&d_ad, 4);
  hipMemcpyAsync(d_off, off, 16, hipMemcpyHostToDevice, cx.stream);
  hipLaunchKernelGGL(k_adler32, dim3(1), dim3(kAdlerThreads), 0, cx.stream, d_bytes,
                     (const uint64_t *)d_off, (uint64_t)0, (const uint32_t *)nullptr, d_ad);
  hipMemcpyAsync(&d->adler, d_ad, 4, hipMemcpyDeviceToHost, cx.stream);
  hipError_t e = hipStreamSynchronize(cx.stream);
  hipFree(d_off);
  hipFree(d_ad);
  if (e != hipSuccess) {
    ldh_dict_free(d);
    fail(LDH_E_HIP, "dict upload: %s", hipGetErrorString(e));
    return nullptr;
  }
  return d;
}
void ldh_dict_free(ldh_dict *d) {
  if (!d) return;
  for (auto &e : d->dev) {  // (work that still reads a copy must have been synchronised by the caller)
    hipSetDevice(e.first);
    hipDeviceSynchronize();
    hipFree(e.second);
  }
  delete d;
}
uint32_t ldh_dict_adler32(const ldh_dict *d) { return d ? d->adler : 0; }
size_t ldh_dict_len(const ldh_dict *d) { return d ? d->bytes.size() : 0; }

// ---- caller-pinned host memory -----------------------------------------------------
int ldh_host_register(void *p, size_t bytes) {
  Ctx *c = current();
  if (!c) return fail(LDH_E_NOT_INIT, "ldh_init() has not been called");
  if (!p || !bytes) return fail(LDH_E_ARG, "ldh_host_register: empty range");
  HIPCHK(hipSetDevice(c->device));
  // pages are what gets pinned: the whole pages the range touches (hipHostRegister wants page granularity)
  const uintptr_t pg = 4096, lo = (uintptr_t)p & ~(pg - 1), hi = ((uintptr_t)p + bytes + pg - 1) & ~(pg - 1);
  const hipError_t e = hipHostRegister((void *)lo, hi - lo, hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(LDH_E_HIP, "hipHostRegister(%p, %zu): %s", (void *)lo, (size_t)(hi - lo), hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lk(g_host_mu);
  g_host_ranges.push_back({(uintptr_t)p, (uintptr_t)p + bytes, false});
  return 0;
}
int ldh_host_unregister(void *p) {
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t i = 0;
    for (; i < g_host_ranges.size(); i++)
      if (g_host_ranges[i].lo == (uintptr_t)p && !g_host_ranges[i].owned) break;
    if (i == g_host_ranges.size()) return fail(LDH_E_ARG, "ldh_host_unregister: %p was not registered", p);
    g_host_ranges.erase(g_host_ranges.begin() + (long)i);
  }
  HIPCHK(hipHostUnregister((void *)((uintptr_t)p & ~(uintptr_t)4095)));
  return 0;
}
void *ldh_host_alloc(size_t bytes) {
  Ctx *c = current();
  if (!c) {
    fail(LDH_E_NOT_INIT, "ldh_init() has not been called");
    return nullptr;
  }
  void *p = nullptr;
  g_allocs++;
  if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    fail(LDH_E_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_host_mu);
  g_host_ranges.push_back({(uintptr_t)p, (uintptr_t)p + (bytes ? bytes : 1), true});
  return p;
}
void ldh_host_free(void *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (size_t i = 0; i < g_host_ranges.size(); i++)
      if (g_host_ranges[i].lo == (uintptr_t)p && g_host_ranges[i].owned) {
        g_host_ranges.erase(g_host_ranges.begin() + (long)i);
        break;
      }
  }
  hipHostFree(p);
}

// ---- batched, device pointers -------------------------------------------------
int ldh_compress_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count, size_t max_in_len,
                           int fmt, int level, int strategy, const ldh_dict *dict, uint8_t *d_out,
                           size_t out_stride, uint32_t *d_out_len, uint8_t *d_pad_bits) {
  CTX_ENTER(cx);
  return compress_dev_locked(cx, d_in, d_in_off, count, max_in_len, fmt, level, strategy, dict, d_out, out_stride,
                             d_out_len, d_pad_bits);
}

int ldh_decompress_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count, int fmt,
                             const ldh_dict *dict, uint8_t *d_out, size_t out_stride, uint32_t *d_out_len,
                             uint32_t *d_bytes_left, int32_t *d_status) {
  CTX_ENTER(cx);
  return decompress_dev_locked(cx, d_in, d_in_off, nullptr, 0, count, fmt, dict, d_out, out_stride, d_out_len,
                               d_bytes_left, d_status);
}

int ldh_decompress_slots_dev(const uint8_t *d_in, size_t in_stride, const uint32_t *d_in_len, size_t count,
                             int fmt, const ldh_dict *dict, uint8_t *d_out, size_t out_stride,
                             uint32_t *d_out_len, uint32_t *d_bytes_left, int32_t *d_status) {
  CTX_ENTER(cx);
  if (!d_in_len) return fail(LDH_E_ARG, "d_in_len is required");
  return decompress_dev_locked(cx, d_in, nullptr, d_in_len, in_stride, count, fmt, dict, d_out, out_stride, d_out_len,
                               d_bytes_left, d_status);
}

// The one entry point that allocates nothing (callers under HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc): ldh_decompress_batch_dev with its two per-stream scratch arrays (zlib trailer values, Adler-32
// sums) in this GPU's copy of the file-scope `scratch` instead of grow-only workspace.  Every stream's entries are
// written before they are read (k_inflate and k_adler32 write all `count` of them, k_zlib_check reads them), so
// nothing has to be cleared between calls; calls that overlap on one GPU would share the array and race (the
// context mutex serialises the enqueue, the library stream the execution).
int ldh_decompress_batch_dev_noalloc(const uint8_t *d_in, const uint64_t *d_in_off, size_t count, int fmt,
                                     const ldh_dict *dict, uint8_t *d_out, size_t out_stride, uint32_t *d_out_len,
                                     uint32_t *d_bytes_left, int32_t *d_status) {
  CTX_ENTER(cx);
  if (count > kNoAllocMaxStreams)
    return fail(LDH_E_CAPACITY, "ldh_decompress_batch_dev_noalloc takes at most %zu streams per call (got %zu)",
                kNoAllocMaxStreams, count);
  if (dict) {  // the dictionary must already be on this GPU (ldh_dict_create on this context, or an earlier call)
    ldh_dict *d = const_cast<ldh_dict *>(dict);
    std::lock_guard<std::mutex> dl(d->mu);
    bool here = false;
    for (auto &e : d->dev) here = here || e.first == cx.device;
    if (!here) return fail(LDH_E_ARG, "dictionary is not resident on device %d (uploading it would allocate)", cx.device);
  }
  const bool prof = cx.prof;  // (profiling brackets launches with events from a pool that may have to grow)
  cx.prof = false;
  int rc = 0;
  {
    // the device's one scratch array: behind the last no-allocation call another context made on this GPU
    std::lock_guard<std::mutex> sl(g_scratch_mu);
    std::pair<int, Ctx *> *slot = nullptr;
    for (auto &e : g_scratch_last)
      if (e.first == cx.device) slot = &e;
    if (!slot) {  // (first such call on this device: one table entry per GPU, reserved at start-up below)
      g_scratch_last.push_back({cx.device, nullptr});
      slot = &g_scratch_last.back();
    }
    if (slot->second && slot->second != &cx) {
      const hipError_t e = hipStreamWaitEvent(cx.stream, slot->second->ev_noalloc, 0);
      if (e != hipSuccess) rc = fail(LDH_E_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e));
    }
    if (!rc)
      rc = decompress_dev_locked(cx, d_in, d_in_off, nullptr, 0, count, fmt, dict, d_out, out_stride, d_out_len,
                                 d_bytes_left, d_status, cx.noalloc);
    if (!rc) {
      const hipError_t e = hipEventRecord(cx.ev_noalloc, cx.stream);
      if (e != hipSuccess) rc = fail(LDH_E_HIP, "hipEventRecord: %s", hipGetErrorString(e));
      slot->second = &cx;
    }
  }
  cx.prof = prof;
  return rc;
}

int ldh_adler32_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count, uint32_t *d_adler) {
  CTX_ENTER(cx);
  if (!count) return 0;
  LAUNCH("k_adler32", k_adler32, (uint32_t)count, kAdlerThreads, 0, d_in, d_in_off, (uint64_t)0,
         (const uint32_t *)nullptr, d_adler);
  HIPCHK(hipGetLastError());
  return 0;
}

int ldh_print6_encode_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count, uint8_t *d_out,
                                size_t out_stride, uint32_t *d_out_len) {
  CTX_ENTER(cx);
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  LAUNCH("k_print6_encode", k_print6_encode, (uint32_t)count, kP6Threads, 0, d_in, d_in_off, d_out,
         (uint64_t)out_stride, d_out_len);
  HIPCHK(hipGetLastError());
  return 0;
}

int ldh_print6_decode_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count, uint8_t *d_out,
                                size_t out_stride, uint32_t *d_out_len, uint8_t *d_ok) {
  CTX_ENTER(cx);
  if (!count) return 0;
  LAUNCH("k_print6_decode", k_print6_decode, (uint32_t)count, kP6Threads, 0, d_in, d_in_off, d_out,
         (uint64_t)out_stride, d_out_len, d_ok);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- batched, host pointers ---------------------------------------------------
static int compress_batch_on(Ctx *on, const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt, int level,
                             int strategy, const ldh_dict *dict, uint8_t *out_base, size_t out_stride,
                             uint32_t *out_len, uint8_t *pad_bits) {
  CTX_ENTER_ON(cx, on);
  if (count == 0) return 0;
  if (level < -1 || level > 9) return fail(LDH_E_ARG, "level must be -1..9 (got %d)", level);
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  if (int rc = pipeline_init(cx)) return rc;
  DrainGuard drain;
  t_near = &cx.near_cpus;
  HostTrace tr;
  // caller-pinned memory (ldh_host_register / ldh_host_alloc) is copied from and to directly
  const bool pin_in = host_pinned(in_base + in_off[0], in_off[count] - in_off[0]);
  const bool pin_out = host_pinned(out_base, count * out_stride);
  size_t npieces = 0;
  // group streams by resolved level (LD:1782-1790) and by size class so that the
  // per-stream workspace (sized by the longest stream of a launch) is not wasted
  struct Group {
    int level;
    std::vector<size_t> idx;
    size_t max_len = 0;
  };
  std::vector<Group> groups;
  for (size_t s = 0; s < count; s++) {
    const size_t n = in_off[s + 1] - in_off[s];
    int lv = level;
    if (lv < 0) lv = n < 2048 ? 7 : (n > 65536 ? 3 : 5);
    int cls = 0;
    while (((size_t)4096 << cls) < n && cls < 20) cls++;
    const int key = lv * 32 + cls;
    Group *g = nullptr;
    for (auto &gg : groups)
      if (gg.level == key) g = &gg;
    if (!g) {
      groups.push_back({key, {}, 0});
      g = &groups.back();
    }
    g->idx.push_back(s);
    g->max_len = std::max(g->max_len, n);
  }
  for (auto &g : groups) {
    const int lv = g.level / 32;
    if (out_stride < ldh_compress_bound(g.max_len))
      return fail(LDH_E_CAPACITY, "out_stride %zu < ldh_compress_bound(%zu)", out_stride, g.max_len);
    // pieces [first, last) of g.idx
    struct Piece {
      size_t first, last;
      uint64_t bytes;
      bool contiguous;  // streams idx[first] .. idx[last-1] are neighbours in the caller's batch
    };
    std::vector<Piece> pieces;
    // (a large batch in pieces of 128 MiB: 1024 streams of 128 KiB let a workgroup of the link and match kernels take a
    // whole stream; 8192 x 128 KiB from pinned memory 24.6 -> 26.1 GB/s)
    static const size_t big_from = getenv("LDH_COMPRESS_BIG_FROM_MB") ? (size_t)atoi(getenv("LDH_COMPRESS_BIG_FROM_MB")) : 768;
    const size_t piece = piece_bytes() * ((!getenv("LDH_PIECE_KB") && in_off[count] - in_off[0] >= (big_from << 20)) ? 2 : 1);
    for (size_t i = 0; i < g.idx.size();) {
      Piece pc{i, i, 0, true};
      while (pc.last < g.idx.size() && (pc.last == pc.first || pc.bytes < piece)) {
        pc.bytes += in_off[g.idx[pc.last] + 1] - in_off[g.idx[pc.last]];
        pc.last++;
      }
      pc.contiguous = g.idx[pc.last - 1] - g.idx[pc.first] + 1 == pc.last - pc.first;
      pieces.push_back(pc);
      i = pc.last;
    }
    npieces += pieces.size();
    auto finish = [&](size_t k) -> int {  // results of piece k to the caller
      Slot &sl = cx.slot[k & 1];
      const Piece &pc = pieces[k];
      const size_t gc = pc.last - pc.first;
      tr.lap(5);
      HIPCHK(hipEventSynchronize(sl.ev_done));
      tr.lap(2);
      const uint32_t *lens = (const uint32_t *)sl.pin_small.p;
      const uint8_t *pads = (const uint8_t *)sl.pin_small.p + gc * 4;
      size_t w = 0;
      for (size_t i = 0; i < gc; i++) w = std::max<size_t>(w, lens[i]);
      w = std::min(round_up(w, 16), out_stride);
      const size_t *idx = g.idx.data() + pc.first;
      if (pin_out && pc.contiguous) {
        // straight into the caller's slots: rows as wide as the piece's longest result
        for (size_t i = 0; i < gc; i++) {
          out_len[idx[i]] = lens[i];
          pad_bits[idx[i]] = pads[i];
        }
        if (w) {
          HIPCHK(hipMemcpy2DAsync(out_base + idx[0] * out_stride, out_stride, sl.d_out.p, out_stride, w, gc,
                                  hipMemcpyDeviceToHost, cx.s_out));
          HIPCHK(hipStreamSynchronize(cx.s_out));
        }
        tr.lap(3);
        return 0;
      }
      if (int rc = ensure_pinned(sl.pin_out, gc * w + 64)) return rc;
      if (w) {
        HIPCHK(hipMemcpy2DAsync(sl.pin_out.p, w, sl.d_out.p, out_stride, w, gc, hipMemcpyDeviceToHost, cx.s_out));
        HIPCHK(hipStreamSynchronize(cx.s_out));
      }
      tr.lap(3);
      const uint8_t *rows = (const uint8_t *)sl.pin_out.p;
      par_ranges(gc, gc * w, [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
          out_len[idx[i]] = lens[i];
          pad_bits[idx[i]] = pads[i];
          memcpy(out_base + idx[i] * out_stride, rows + i * w, lens[i]);
        }
      });
      tr.lap(4);
      return 0;
    };
    for (size_t k = 0; k < pieces.size(); k++) {
      Slot &sl = cx.slot[k & 1];
      const Piece &pc = pieces[k];
      const size_t gc = pc.last - pc.first;
      const bool direct_in = pin_in && pc.contiguous;
      tr.lap(5);
      // (piece k-2, the last user of this slot, was finished before piece k-1 was enqueued)
      if (!direct_in)
        if (int rc = ensure_pinned(sl.pin_in, pc.bytes + 64)) return rc;
      if (int rc = ensure_pinned(sl.pin_off, (gc + 1) * 8)) return rc;
      if (int rc = ensure_pinned(sl.pin_small, gc * 5 + 64)) return rc;
      if (int rc = ensure(sl.d_in, pc.bytes + 64)) return rc;
      if (int rc = ensure(sl.d_off, (gc + 1) * 8)) return rc;
      if (int rc = ensure(sl.d_out, gc * out_stride)) return rc;
      if (int rc = ensure(sl.d_len, gc * 4 + 64)) return rc;
      if (int rc = ensure(sl.d_a8, gc + 64)) return rc;
      uint64_t *off = (uint64_t *)sl.pin_off.p;
      uint64_t tot = 0;
      size_t max_len = 0;
      for (size_t i = 0; i < gc; i++) {
        off[i] = tot;
        const size_t n = in_off[g.idx[pc.first + i] + 1] - in_off[g.idx[pc.first + i]];
        tot += n;
        max_len = std::max(max_len, n);
      }
      off[gc] = tot;
      const size_t *idx = g.idx.data() + pc.first;
      const uint8_t *src = (const uint8_t *)sl.pin_in.p;
      if (direct_in) {
        src = in_base + in_off[idx[0]];
      } else if (pc.contiguous) {
        par_memcpy(sl.pin_in.p, in_base + in_off[idx[0]], tot);
      } else {
        uint8_t *dst = (uint8_t *)sl.pin_in.p;
        par_ranges(gc, tot, [=](size_t lo, size_t hi) {
          for (size_t i = lo; i < hi; i++) memcpy(dst + off[i], in_base + in_off[idx[i]], off[i + 1] - off[i]);
        });
      }
      if (tot) HIPCHK(hipMemcpyAsync(sl.d_in.p, src, tot, hipMemcpyHostToDevice, cx.s_in));
      HIPCHK(hipMemcpyAsync(sl.d_off.p, sl.pin_off.p, (gc + 1) * 8, hipMemcpyHostToDevice, cx.s_in));
      HIPCHK(hipEventRecord(sl.ev_in, cx.s_in));
      HIPCHK(hipStreamWaitEvent(cx.stream, sl.ev_in, 0));
      tr.lap(0);
      if (int rc = compress_dev_locked(cx, (const uint8_t *)sl.d_in.p, (const uint64_t *)sl.d_off.p, gc, max_len, fmt, lv,
                                       strategy, dict, (uint8_t *)sl.d_out.p, out_stride, (uint32_t *)sl.d_len.p,
                                       (uint8_t *)sl.d_a8.p))
        return rc;
      HIPCHK(hipMemcpyAsync(sl.pin_small.p, sl.d_len.p, gc * 4, hipMemcpyDeviceToHost, cx.stream));
      HIPCHK(hipMemcpyAsync((uint8_t *)sl.pin_small.p + gc * 4, sl.d_a8.p, gc, hipMemcpyDeviceToHost, cx.stream));
      HIPCHK(hipEventRecord(sl.ev_done, cx.stream));
      tr.lap(1);
      if (k > 0)
        if (int rc = finish(k - 1)) return rc;
    }
    if (int rc = finish(pieces.size() - 1)) return rc;
  }
  drain.clean = true;
  tr.report("ldh_compress_batch", npieces, pin_in, pin_out);
  return 0;
}

static int decompress_batch_on(Ctx *on, const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt,
                               const ldh_dict *dict, uint8_t *out_base, size_t out_stride, uint32_t *out_len,
                               uint32_t *bytes_left, int32_t *status) {
  CTX_ENTER_ON(cx, on);
  if (count == 0) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  if (int rc = pipeline_init(cx)) return rc;
  DrainGuard drain;
  t_near = &cx.near_cpus;
  HostTrace tr;
  const bool pin_in = host_pinned(in_base + in_off[0], in_off[count] - in_off[0]);
  const bool pin_out = host_pinned(out_base, count * out_stride);
  struct Piece {
    size_t first, last;
  };
  constexpr size_t kInflateSlots = 5, kInflateAhead = 3;  // (a slot is free again two pieces before it is needed)
  static_assert(sizeof(cx.slot) / sizeof(cx.slot[0]) >= kInflateSlots, "slots");
  std::vector<Piece> pieces;
  const size_t piece = inflate_piece_bytes(count * out_stride);
  for (size_t i = 0; i < count;) {
    size_t j = i + 1;
    while (j < count && (j - i + 1) * out_stride <= piece && in_off[j + 1] - in_off[i] <= piece) j++;
    pieces.push_back({i, j});
    i = j;
  }
  auto start_out = [&](size_t k) -> int {  // piece k is done on the device: start copying its results out
    Slot &sl = cx.slot[k % kInflateSlots];
    const Piece &pc = pieces[k];
    const size_t gc = pc.last - pc.first;
    tr.lap(5);
    HIPCHK(hipEventSynchronize(sl.ev_done));
    tr.lap(2);
    const uint32_t *lens = (const uint32_t *)sl.pin_small.p;
    const int32_t *st = (const int32_t *)((const uint8_t *)sl.pin_small.p + gc * 8);
    memcpy(out_len + pc.first, lens, gc * 4);
    memcpy(bytes_left + pc.first, (const uint8_t *)sl.pin_small.p + gc * 4, gc * 4);
    memcpy(status + pc.first, st, gc * 4);
    size_t w = 0;
    for (size_t i = 0; i < gc; i++)
      if (st[i] == 0) w = std::max<size_t>(w, lens[i]);
    w = std::min(round_up(w, 16), out_stride);
    sl.w = w;
    if (!w) return 0;
    if (pin_out) {
      // straight into the caller's slots (rows as wide as the piece's longest result; a slot whose stream failed, or
      // the part of a slot beyond its own result, receives whatever the device buffer held there)
      uint8_t *dst = out_base + pc.first * out_stride;
      if (w == out_stride)
        HIPCHK(hipMemcpyAsync(dst, sl.d_out.p, gc * w, hipMemcpyDeviceToHost, cx.s_out));
      else
        HIPCHK(hipMemcpy2DAsync(dst, out_stride, sl.d_out.p, out_stride, w, gc, hipMemcpyDeviceToHost, cx.s_out));
      HIPCHK(hipEventRecord(sl.ev_out, cx.s_out));
      return 0;
    }
    if (int rc = ensure_pinned(sl.pin_out, gc * w + 64)) return rc;
    if (w == out_stride)
      HIPCHK(hipMemcpyAsync(sl.pin_out.p, sl.d_out.p, gc * w, hipMemcpyDeviceToHost, cx.s_out));
    else
      HIPCHK(hipMemcpy2DAsync(sl.pin_out.p, w, sl.d_out.p, out_stride, w, gc, hipMemcpyDeviceToHost, cx.s_out));
    HIPCHK(hipEventRecord(sl.ev_out, cx.s_out));
    return 0;
  };
  auto scatter = [&](size_t k) -> int {  // results of piece k from pinned memory to the caller's slots
    Slot &sl = cx.slot[k % kInflateSlots];
    const Piece &pc = pieces[k];
    const size_t gc = pc.last - pc.first, w = sl.w;
    if (!w) return 0;
    tr.lap(5);
    HIPCHK(hipEventSynchronize(sl.ev_out));
    tr.lap(3);
    if (pin_out) return 0;
    const uint32_t *lens = (const uint32_t *)sl.pin_small.p;
    const int32_t *st = (const int32_t *)((const uint8_t *)sl.pin_small.p + gc * 8);
    const uint8_t *rows = (const uint8_t *)sl.pin_out.p;
    uint8_t *dst = out_base + pc.first * out_stride;
    par_ranges(gc, gc * w, [=](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++)
        if (st[i] == 0 && lens[i]) memcpy(dst + i * out_stride, rows + i * w, lens[i]);
    });
    tr.lap(4);
    return 0;
  };
  for (size_t k = 0; k < pieces.size(); k++) {
    Slot &sl = cx.slot[k % kInflateSlots];  // (its last user, piece k-5, was scattered before piece k-1 was enqueued)
    const Piece &pc = pieces[k];
    const size_t gc = pc.last - pc.first;
    const uint64_t base = in_off[pc.first], tot = in_off[pc.last] - base;
    tr.lap(5);
    if (!pin_in)
      if (int rc = ensure_pinned(sl.pin_in, tot + 64)) return rc;
    if (int rc = ensure_pinned(sl.pin_off, (gc + 1) * 8)) return rc;
    if (int rc = ensure_pinned(sl.pin_small, gc * 12 + 64)) return rc;
    if (int rc = ensure(sl.d_in, tot + 64)) return rc;
    if (int rc = ensure(sl.d_off, (gc + 1) * 8)) return rc;
    if (int rc = ensure(sl.d_out, gc * out_stride + 64)) return rc;
    if (int rc = ensure(sl.d_len, gc * 4 + 64)) return rc;
    if (int rc = ensure(sl.d_a32a, gc * 4 + 64)) return rc;
    if (int rc = ensure(sl.d_a32b, gc * 4 + 64)) return rc;
    // offsets are re-based to start at 0 so that the device copy of the data is aligned
    uint64_t *off = (uint64_t *)sl.pin_off.p;
    for (size_t i = 0; i <= gc; i++) off[i] = in_off[pc.first + i] - base;
    const uint8_t *src = in_base + base;
    if (!pin_in) {
      par_memcpy(sl.pin_in.p, in_base + base, tot);
      src = (const uint8_t *)sl.pin_in.p;
    }
    if (tot) HIPCHK(hipMemcpyAsync(sl.d_in.p, src, tot, hipMemcpyHostToDevice, cx.s_in));
    HIPCHK(hipMemcpyAsync(sl.d_off.p, sl.pin_off.p, (gc + 1) * 8, hipMemcpyHostToDevice, cx.s_in));
    HIPCHK(hipEventRecord(sl.ev_in, cx.s_in));
    // Neighbouring pieces' kernels run side by side, on two streams, one wavefront per stream: a piece of one or two
    // thousand streams is bound by what one stream's decoder waits for, not by what the chip can issue (2048 streams
    // take 3.5 ms, 8192 take 6.4), so two pieces in flight finish in little more than the time of one -- and the
    // first one's results are on their way to the host that much earlier.  (The wide kernels fill the register
    // files with one piece; they are for a call that is a single piece.)
    const bool many = pieces.size() > 1;
    StreamSwap on(cx, many && (k & 1) ? cx.s_k2 : cx.stream);
    HIPCHK(hipStreamWaitEvent(cx.stream, sl.ev_in, 0));
    tr.lap(0);
    if (int rc = ensure(sl.d_scr, gc * 8 + 64)) return rc;
    if (int rc = decompress_dev_locked(cx, (const uint8_t *)sl.d_in.p, (const uint64_t *)sl.d_off.p, nullptr, 0, gc, fmt,
                                       dict, (uint8_t *)sl.d_out.p, out_stride, (uint32_t *)sl.d_len.p,
                                       (uint32_t *)sl.d_a32a.p, (int32_t *)sl.d_a32b.p, (uint32_t *)sl.d_scr.p,
                                       (uint32_t *)sl.d_scr.p + gc, many ? 1 : 0))
      return rc;
    uint8_t *small = (uint8_t *)sl.pin_small.p;
    HIPCHK(hipMemcpyAsync(small, sl.d_len.p, gc * 4, hipMemcpyDeviceToHost, cx.stream));
    HIPCHK(hipMemcpyAsync(small + gc * 4, sl.d_a32a.p, gc * 4, hipMemcpyDeviceToHost, cx.stream));
    HIPCHK(hipMemcpyAsync(small + gc * 8, sl.d_a32b.p, gc * 4, hipMemcpyDeviceToHost, cx.stream));
    HIPCHK(hipEventRecord(sl.ev_done, cx.stream));
    on.restore();
    tr.lap(1);
    // (three pieces are enqueued before the first one's results are waited for: with two kernel streams a piece's
    // input must already be on the device when the stream it will run on falls free)
    if (k + 1 >= kInflateAhead)
      if (int rc = start_out(k + 1 - kInflateAhead)) return rc;
    if (k >= kInflateAhead)
      if (int rc = scatter(k - kInflateAhead)) return rc;
  }
  const size_t np = pieces.size();
  for (size_t k = np >= kInflateAhead ? np + 1 - kInflateAhead : 0; k < np; k++)
    if (int rc = start_out(k)) return rc;
  for (size_t k = np > kInflateAhead ? np - kInflateAhead : 0; k < np; k++)
    if (int rc = scatter(k)) return rc;
  drain.clean = true;
  tr.report("ldh_decompress_batch", np, pin_in, pin_out);
  return 0;
}

// The contexts a host-pointer batch call runs on: the pool of ldh_init_devices() (unless this thread chose one
// of them with ldh_use_context), else the one default context.
static std::vector<Ctx *> batch_contexts() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (t_cur || g_pool.size() < 2) return {t_cur ? t_cur : g_default};
  return g_pool;
}
// f(context, first, last) for the byte-balanced contiguous part of the batch each context takes (host-side
// scatter, one host thread per context, no inter-GPU exchange; results land in the caller's one buffer)
extern "C++" template <class F>
static int over_contexts(const uint64_t *in_off, size_t count, F f) {
  const std::vector<Ctx *> cxs = batch_contexts();
  if (cxs.size() < 2 || count < 2) return f(cxs[0], (size_t)0, count);
  const int P = (int)cxs.size();
  std::vector<size_t> first(P + 1);
  if (int rc = ldh_partition_by_bytes(in_off, count, P, first.data())) return rc;
  std::vector<int> rcs(P, 0);
  std::vector<std::string> errs(P);
  std::vector<std::thread> th;
  for (int p = 0; p < P; p++)
    th.emplace_back([&, p] {
      if (first[p + 1] == first[p]) return;
      bind_this_thread(cxs[p]->near_cpus);  // this context's host thread runs beside its GPU
      rcs[p] = f(cxs[p], first[p], first[p + 1]);
      if (rcs[p]) errs[p] = g_err;  // (thread-local: hand it to the caller's thread)
    });
  for (auto &t : th) t.join();
  for (int p = 0; p < P; p++)
    if (rcs[p]) return fail(rcs[p], "context %d (device %d): %s", p, cxs[p]->device, errs[p].c_str());
  return 0;
}

int ldh_compress_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt, int level,
                       int strategy, const ldh_dict *dict, uint8_t *out_base, size_t out_stride,
                       uint32_t *out_len, uint8_t *pad_bits) {
  return over_contexts(in_off, count, [=](Ctx *c, size_t a, size_t b) {
    return compress_batch_on(c, in_base, in_off + a, b - a, fmt, level, strategy, dict, out_base + a * out_stride,
                             out_stride, out_len + a, pad_bits + a);
  });
}

int ldh_decompress_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt,
                         const ldh_dict *dict, uint8_t *out_base, size_t out_stride, uint32_t *out_len,
                         uint32_t *bytes_left, int32_t *status) {
  return over_contexts(in_off, count, [=](Ctx *c, size_t a, size_t b) {
    return decompress_batch_on(c, in_base, in_off + a, b - a, fmt, dict, out_base + a * out_stride, out_stride,
                               out_len + a, bytes_left + a, status + a);
  });
}

int ldh_adler32_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, uint32_t *adler) {
  CTX_ENTER(cx);
  if (!count) return 0;
  if (int rc = upload_inputs(cx, in_base, in_off, count)) return rc;
  if (int rc = ensure(cx.h_len, count * 4 + 64)) return rc;
  LAUNCH("k_adler32", k_adler32, (uint32_t)count, kAdlerThreads, 0, (const uint8_t *)cx.h_in.p,
         (const uint64_t *)cx.h_off.p, (uint64_t)0, (const uint32_t *)nullptr, (uint32_t *)cx.h_len.p);
  HIPCHK(hipMemcpyAsync(adler, cx.h_len.p, count * 4, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));
  return 0;
}

int ldh_print6_encode_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, uint8_t *out_base,
                            size_t out_stride, uint32_t *out_len) {
  CTX_ENTER(cx);
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  for (size_t s = 0; s < count; s++)
    if (((in_off[s + 1] - in_off[s]) * 8 + 5) / 6 > out_stride)
      return fail(LDH_E_CAPACITY, "out_stride too small for stream %zu", s);
  if (int rc = upload_inputs(cx, in_base, in_off, count)) return rc;
  if (int rc = ensure(cx.h_out, count * out_stride + 64)) return rc;
  if (int rc = ensure(cx.h_len, count * 4 + 64)) return rc;
  LAUNCH("k_print6_encode", k_print6_encode, (uint32_t)count, kP6Threads, 0, (const uint8_t *)cx.h_in.p,
         (const uint64_t *)cx.h_off.p, (uint8_t *)cx.h_out.p, (uint64_t)out_stride, (uint32_t *)cx.h_len.p);
  HIPCHK(hipMemcpyAsync(out_len, cx.h_len.p, count * 4, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipMemcpy2DAsync(out_base, out_stride, cx.h_out.p, out_stride, out_stride, count, hipMemcpyDeviceToHost,
                          cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));
  return 0;
}

int ldh_print6_decode_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, uint8_t *out_base,
                            size_t out_stride, uint32_t *out_len, uint8_t *ok) {
  CTX_ENTER(cx);
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  for (size_t s = 0; s < count; s++)
    if ((in_off[s + 1] - in_off[s]) * 3 / 4 + 3 > out_stride)
      return fail(LDH_E_CAPACITY, "out_stride too small for stream %zu", s);
  if (int rc = upload_inputs(cx, in_base, in_off, count)) return rc;
  if (int rc = ensure(cx.h_out, count * out_stride + 64)) return rc;
  if (int rc = ensure(cx.h_len, count * 4 + 64)) return rc;
  if (int rc = ensure(cx.h_aux8, count + 64)) return rc;
  LAUNCH("k_print6_decode", k_print6_decode, (uint32_t)count, kP6Threads, 0, (const uint8_t *)cx.h_in.p,
         (const uint64_t *)cx.h_off.p, (uint8_t *)cx.h_out.p, (uint64_t)out_stride, (uint32_t *)cx.h_len.p,
         (uint8_t *)cx.h_aux8.p);
  HIPCHK(hipMemcpyAsync(out_len, cx.h_len.p, count * 4, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipMemcpyAsync(ok, cx.h_aux8.p, count, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipMemcpy2DAsync(out_base, out_stride, cx.h_out.p, out_stride, out_stride, count, hipMemcpyDeviceToHost,
                          cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));
  return 0;
}

// ---- single-string forms ------------------------------------------------------
int ldh_compress(const uint8_t *in, size_t n, int fmt, int level, int strategy, const ldh_dict *dict,
                 uint8_t *out, size_t cap, size_t *out_len, int *pad_bits) {
  const uint64_t off[2] = {0, n};
  const size_t stride = ldh_compress_bound(n);
  std::vector<uint8_t> tmp(stride);
  uint32_t ol = 0;
  uint8_t pb = 0;
  static const uint8_t empty = 0;
  int rc = ldh_compress_batch(in ? in : &empty, off, 1, fmt, level, strategy, dict, tmp.data(), stride, &ol, &pb);
  if (rc) return rc;
  if (out_len) *out_len = ol;
  if (pad_bits) *pad_bits = pb;
  if (ol > cap) return fail(LDH_E_CAPACITY, "output needs %u bytes, cap is %zu", ol, cap);
  memcpy(out, tmp.data(), ol);
  return 0;
}

int ldh_decompress(const uint8_t *in, size_t n, int fmt, const ldh_dict *dict, uint8_t *out, size_t cap,
                   size_t *out_len, size_t *bytes_left, int *status) {
  const uint64_t off[2] = {0, n};
  const size_t stride = round_up(std::max<size_t>(cap, 16), 16);
  std::vector<uint8_t> tmp(stride);
  uint32_t ol = 0, bl = 0;
  int32_t st = 0;
  static const uint8_t empty = 0;
  int rc = ldh_decompress_batch(in ? in : &empty, off, 1, fmt, dict, tmp.data(), stride, &ol, &bl, &st);
  if (rc) return rc;
  if (st == 0 && ol > cap) st = LDH_STATUS_OUTPUT_FULL;
  if (status) *status = st;
  if (out_len) *out_len = ol;
  if (bytes_left) *bytes_left = bl;
  if (st == 0 && ol) memcpy(out, tmp.data(), ol);
  return 0;
}

int ldh_adler32(const uint8_t *s, size_t n, uint32_t *adler) {
  const uint64_t off[2] = {0, n};
  static const uint8_t empty = 0;
  return ldh_adler32_batch(s ? s : &empty, off, 1, adler);
}

// ---- CreateCodec codecs (LD:2921-3158) ------------------------------------------
ldh_codec *ldh_codec_create(const uint8_t *reserved, size_t nr, const uint8_t *escape, size_t ne,
                            const uint8_t *map, size_t nm, const char **err) {
  static const char *kDup =
      "There must be no duplicate characters in the concatenation of reserved_chars, escape_chars and map_chars.";
  const char *dummy;
  if (!err) err = &dummy;
  *err = nullptr;
  if (ne == 0) { *err = "No escape characters supplied."; return nullptr; }  // LD:2928
  if (nr < nm) {                                                             // LD:2929-2932
    *err = "The number of reserved characters must be at least as many as the number of mapped chars.";
    return nullptr;
  }
  if (nr == 0) { *err = "No characters to encode."; return nullptr; }        // LD:2933
  std::vector<uint8_t> all;
  all.insert(all.end(), reserved, reserved + nr);
  all.insert(all.end(), escape, escape + ne);
  all.insert(all.end(), map, map + nm);
  bool taken[256] = {false}, done[256] = {false};
  for (uint8_t b : all) {  // LD:2937-2946
    if (taken[b]) { *err = kDup; return nullptr; }
    taken[b] = true;
  }
  // host image: header, then one suffix table per escape character that gets used
  std::vector<uint16_t> dec(256, 0xFFFF);
  std::vector<std::vector<uint16_t>> passes;
  ldh::CodecTables H;
  memset(&H, 0, sizeof(H));
  for (int b = 0; b < 256; b++) {
    H.enc_len[b] = 1;
    H.enc0[b] = (uint8_t)b;
    H.map_back[b] = (uint8_t)b;
  }
  for (size_t i = 0; i < nr; i++) H.reserved[reserved[i]] = 1;  // LD:3040
  for (size_t i = 0; i < nm; i++) {                             // LD:2957-2971
    H.enc0[reserved[i]] = map[i];
    done[reserved[i]] = true;
    H.map_back[map[i]] = reserved[i];
  }
  size_t esc = 0;
  int r = 0;
  for (size_t i = 0; i < all.size(); i++) {  // LD:2981-3022
    const uint8_t ch = all[i];
    if (!done[ch]) {
      while (r >= 256 || taken[r]) {
        r++;
        if (r > 255) {  // this escape character's suffixes are used up
          H.pass_escape[passes.size()] = escape[esc];
          passes.push_back(dec);
          dec.assign(256, 0xFFFF);
          esc++;
          r = 0;
          if (esc >= ne) { *err = "Out of escape characters."; return nullptr; }  // LD:2999-3005
        }
      }
      H.enc_len[ch] = 2;
      H.enc0[ch] = escape[esc];
      H.enc1[ch] = (uint8_t)r;
      done[ch] = true;
      dec[r] = ch;
      r++;
    }
    if (i + 1 == all.size()) {
      H.pass_escape[passes.size()] = escape[esc];
      passes.push_back(dec);
    }
  }
  H.npass = (uint32_t)passes.size();

  Ctx *cxp = current();
  if (!cxp) { *err = "libdeflate_hip is not initialised"; return nullptr; }
  const size_t head = offsetof(ldh::CodecTables, pass_dec);
  const size_t bytes = head + passes.size() * 512;
  ldh_codec *c = new ldh_codec;
  c->npass = H.npass;
  c->img.resize(bytes);
  memcpy(c->img.data(), &H, head);
  for (size_t k = 0; k < passes.size(); k++) memcpy(c->img.data() + head + k * 512, passes[k].data(), 512);
  {
    std::lock_guard<std::mutex> lk(cxp->mu);
    if (hipSetDevice(cxp->device) != hipSuccess || !codec_on(*cxp, c)) {
      delete c;
      *err = "hipMalloc/hipMemcpy of the codec tables failed";
      fail(LDH_E_NOMEM, "%s", *err);
      return nullptr;
    }
  }
  return c;
}

void ldh_codec_free(ldh_codec *c) {
  if (!c) return;
  for (auto &e : c->dev) {
    hipSetDevice(e.first);
    hipDeviceSynchronize();
    hipFree(e.second);
  }
  delete c;
}

namespace {
int codec_decode_dev_locked(Ctx &cx, const ldh_codec *c, const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                            uint8_t *d_out, size_t out_stride, uint32_t *d_out_len, uint8_t *d_ok) {
  // one launch per gsub pass, ping-pong between the caller's slots and a scratch copy of them so
  // that the last pass lands in the caller's
  if (int rc = ensure(cx.codec_tmp, count * out_stride + 64)) return rc;
  if (int rc = ensure(cx.misc32a, count * 4 + 64)) return rc;
  const CodecTables *d_tab = codec_on(cx, c);
  if (!d_tab) return fail(LDH_E_NOMEM, "codec upload to device %d failed", cx.device);
  uint8_t *tmp = (uint8_t *)cx.codec_tmp.p;
  uint32_t *tmp_len = (uint32_t *)cx.misc32a.p;
  const uint8_t *prev = nullptr;
  const uint32_t *prev_len = nullptr;
  for (uint32_t p = 0; p < c->npass; p++) {
    const bool to_out = ((c->npass - 1 - p) & 1) == 0;
    uint8_t *dst = to_out ? d_out : tmp;
    uint32_t *dst_len = to_out ? d_out_len : tmp_len;
    LAUNCH("k_codec_decode_pass", k_codec_decode_pass, (uint32_t)count, kCodecThreads, 0, d_in, d_in_off, prev,
           prev_len, (uint64_t)out_stride, d_tab, p, dst, dst_len, d_ok);
    prev = dst;
    prev_len = dst_len;
  }
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace

int ldh_codec_encode_batch_dev(const ldh_codec *c, const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                               uint8_t *d_out, size_t out_stride, uint32_t *d_out_len) {
  CTX_ENTER(cx);
  if (!c) return fail(LDH_E_ARG, "codec is null");
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  const CodecTables *d_tab = codec_on(cx, c);
  if (!d_tab) return fail(LDH_E_NOMEM, "codec upload to device %d failed", cx.device);
  LAUNCH("k_codec_encode", k_codec_encode, (uint32_t)count, kCodecThreads, 0, d_in, d_in_off,
         d_tab, d_out, (uint64_t)out_stride, d_out_len);
  HIPCHK(hipGetLastError());
  return 0;
}

int ldh_codec_decode_batch_dev(const ldh_codec *c, const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                               uint8_t *d_out, size_t out_stride, uint32_t *d_out_len, uint8_t *d_ok) {
  CTX_ENTER(cx);
  if (!c) return fail(LDH_E_ARG, "codec is null");
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  return codec_decode_dev_locked(cx, c, d_in, d_in_off, count, d_out, out_stride, d_out_len, d_ok);
}

int ldh_codec_encode_batch(const ldh_codec *c, const uint8_t *in_base, const uint64_t *in_off, size_t count,
                           uint8_t *out_base, size_t out_stride, uint32_t *out_len) {
  CTX_ENTER(cx);
  if (!c) return fail(LDH_E_ARG, "codec is null");
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  for (size_t s = 0; s < count; s++)
    if ((in_off[s + 1] - in_off[s]) * 2 > out_stride)
      return fail(LDH_E_CAPACITY, "out_stride too small for stream %zu", s);
  if (int rc = upload_inputs(cx, in_base, in_off, count)) return rc;
  if (int rc = ensure(cx.h_out, count * out_stride + 64)) return rc;
  if (int rc = ensure(cx.h_len, count * 4 + 64)) return rc;
  const CodecTables *d_tab = codec_on(cx, c);
  if (!d_tab) return fail(LDH_E_NOMEM, "codec upload to device %d failed", cx.device);
  LAUNCH("k_codec_encode", k_codec_encode, (uint32_t)count, kCodecThreads, 0, (const uint8_t *)cx.h_in.p,
         (const uint64_t *)cx.h_off.p, d_tab, (uint8_t *)cx.h_out.p, (uint64_t)out_stride,
         (uint32_t *)cx.h_len.p);
  HIPCHK(hipMemcpyAsync(out_len, cx.h_len.p, count * 4, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipMemcpy2DAsync(out_base, out_stride, cx.h_out.p, out_stride, out_stride, count, hipMemcpyDeviceToHost,
                          cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));
  return 0;
}

int ldh_codec_decode_batch(const ldh_codec *c, const uint8_t *in_base, const uint64_t *in_off, size_t count,
                           uint8_t *out_base, size_t out_stride, uint32_t *out_len, uint8_t *ok) {
  CTX_ENTER(cx);
  if (!c) return fail(LDH_E_ARG, "codec is null");
  if (!count) return 0;
  if (out_stride % 16 != 0) return fail(LDH_E_ARG, "out_stride must be a multiple of 16");
  for (size_t s = 0; s < count; s++)
    if (in_off[s + 1] - in_off[s] > out_stride)
      return fail(LDH_E_CAPACITY, "out_stride too small for stream %zu", s);
  if (int rc = upload_inputs(cx, in_base, in_off, count)) return rc;
  if (int rc = ensure(cx.h_out, count * out_stride + 64)) return rc;
  if (int rc = ensure(cx.h_len, count * 4 + 64)) return rc;
  if (int rc = ensure(cx.h_aux8, count + 64)) return rc;
  if (int rc = codec_decode_dev_locked(cx, c, (const uint8_t *)cx.h_in.p, (const uint64_t *)cx.h_off.p, count,
                                       (uint8_t *)cx.h_out.p, out_stride, (uint32_t *)cx.h_len.p,
                                       (uint8_t *)cx.h_aux8.p))
    return rc;
  HIPCHK(hipMemcpyAsync(out_len, cx.h_len.p, count * 4, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipMemcpyAsync(ok, cx.h_aux8.p, count, hipMemcpyDeviceToHost, cx.stream));
  HIPCHK(hipMemcpy2DAsync(out_base, out_stride, cx.h_out.p, out_stride, out_stride, count, hipMemcpyDeviceToHost,
                          cx.stream));
  HIPCHK(hipStreamSynchronize(cx.stream));
  return 0;
}

int ldh_print6_encode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
  const uint64_t off[2] = {0, n};
  const size_t stride = round_up((n * 8 + 5) / 6 + 16, 16);
  std::vector<uint8_t> tmp(stride);
  uint32_t ol = 0;
  static const uint8_t empty = 0;
  int rc = ldh_print6_encode_batch(in ? in : &empty, off, 1, tmp.data(), stride, &ol);
  if (rc) return rc;
  if (out_len) *out_len = ol;
  if (ol > cap) return fail(LDH_E_CAPACITY, "output needs %u bytes, cap is %zu", ol, cap);
  memcpy(out, tmp.data(), ol);
  return 0;
}

int ldh_print6_decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, int *ok) {
  const uint64_t off[2] = {0, n};
  const size_t stride = round_up(n * 3 / 4 + 16, 16);
  std::vector<uint8_t> tmp(stride);
  uint32_t ol = 0;
  uint8_t k = 0;
  static const uint8_t empty = 0;
  int rc = ldh_print6_decode_batch(in ? in : &empty, off, 1, tmp.data(), stride, &ol, &k);
  if (rc) return rc;
  if (ok) *ok = k;
  if (out_len) *out_len = ol;
  if (k && ol > cap) return fail(LDH_E_CAPACITY, "output needs %u bytes, cap is %zu", ol, cap);
  if (k) memcpy(out, tmp.data(), ol);
  return 0;
}

// ---- measurement -----------------------------------------------------------------
int ldh_profile_enable(int on) {
  CTX_ENTER(cx);
  prof_drain(cx);
  cx.prof = on != 0;
  return 0;
}
int ldh_profile_reset(void) {
  CTX_ENTER(cx);
  prof_drain(cx);
  cx.prof_entries.clear();
  return 0;
}
int ldh_profile_count(void) {
  CTX_ENTER(cx);
  prof_drain(cx);
  return (int)cx.prof_entries.size();
}
int ldh_profile_get(int index, char *name, size_t name_cap, double *ms, uint64_t *launches) {
  CTX_ENTER(cx);
  prof_drain(cx);
  if (index < 0 || index >= (int)cx.prof_entries.size()) return fail(LDH_E_ARG, "bad profile index");
  if (name && name_cap) {
    strncpy(name, cx.prof_entries[index].name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (ms) *ms = cx.prof_entries[index].ms;
  if (launches) *launches = cx.prof_entries[index].launches;
  return 0;
}
int ldh_timer_start(void) {
  CTX_ENTER(cx);
  HIPCHK(hipEventRecord(cx.t0, cx.stream));
  return 0;
}
int ldh_timer_stop(double *ms) {
  CTX_ENTER(cx);
  HIPCHK(hipEventRecord(cx.t1, cx.stream));
  HIPCHK(hipEventSynchronize(cx.t1));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, cx.t0, cx.t1));
  if (ms) *ms = f;
  return 0;
}

// ---- inspection ---------------------------------------------------------------------
uint64_t ldh_debug_alloc_count(void) { return g_allocs.load(); }
#ifdef LDH_WIDE_STAMPS  // diagnostic builds only (not declared in the header)
int ldh_debug_wide_stamps(unsigned long long *dst, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(ldh::g_wide_stamps), 16 * 8);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(ldh::g_wide_stamps), z, 16 * 8);
  }
  return 0;
}
#endif
#ifdef LDH_PARSE_STAMPS  // diagnostic builds only (not declared in the header)
int ldh_debug_parse_stamps(unsigned long long *dst, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(ldh::g_parse_stamps), 16 * 8);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(ldh::g_parse_stamps), z, 16 * 8);
  }
  return 0;
}
#endif
int ldh_debug_fetch(int what, size_t stream, void *dst, size_t cap_bytes, size_t *got_bytes) {
  CTX_ENTER(cx);
  if (!cx.have_last || stream >= cx.last.count) return fail(LDH_E_ARG, "no such stream in the last batch");
  const Batch &B = cx.last;
  const void *src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case 0: src = B.prevd + stream * B.nvp; bytes = 2ull * B.nvp; break;
    case 1: src = B.bufB + stream * B.nvp; bytes = 4ull * B.nvp; break;
    case 2: src = B.bufA + stream * B.nvp; bytes = 4ull * B.nvp; break;
    case 3: src = B.blocks + stream * B.max_blocks; bytes = (size_t)B.max_blocks * sizeof(BlockInfo); break;
    case 4: src = B.vbuf + stream * (size_t)(B.nvp + kVPad); bytes = B.nvp + kVPad; break;
    default: return fail(LDH_E_ARG, "bad what");
  }
  if (bytes > cap_bytes) bytes = cap_bytes;
  HIPCHK(hipStreamSynchronize(cx.stream));
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  if (got_bytes) *got_bytes = bytes;
  return 0;
}

}  // extern "C"
