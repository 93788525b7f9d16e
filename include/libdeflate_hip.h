/*
 * libdeflate_hip.h -- C ABI of libdeflate_hip.so, the MI355X (gfx950) engine behind
 * LibDeflate's compress / decompress hot path.
 *
 * Each entry point names the reference interface it replaces (LD = LibDeflate.lua,
 * upstream v1.0.2-release).  The reference has no FFI seam of its own; the seam is cut
 * at the *Internal functions that its public methods call after argument validation
 * (argument validation and error strings stay in the host-language shim, see
 * INTEGRATION.md).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Return value of every int function: 0 = ok, negative = API failure (LDH_E_*; text in
 * ldh_last_error()).  Per-stream *data* results (LibDeflate's inflate status codes,
 * SURVEY Q11) are written to status[] and never make the call itself fail.
 *
 * Threading: the library keeps one HIP stream and one workspace per context (ldh_init: one
 * context; ldh_init_devices: one per GPU); calls on a context are serialised by its mutex.  Host-pointer entry points synchronise before they
 * return.  *_dev entry points only enqueue work on the library stream (ldh_stream());
 * call ldh_sync() before reading results on the host.
 */
#ifndef LIBDEFLATE_HIP_H
#define LIBDEFLATE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDH_FMT_DEFLATE 0 /* CompressDeflate / DecompressDeflate */
#define LDH_FMT_ZLIB 1    /* CompressZlib / DecompressZlib */

/* configs.strategy (LD:704-705) */
#define LDH_STRATEGY_DEFAULT 0
#define LDH_STRATEGY_FIXED 1
#define LDH_STRATEGY_HUFFMAN_ONLY 2
#define LDH_STRATEGY_DYNAMIC 3

#define LDH_LEVEL_DEFAULT (-1) /* by input size, LD:1782-1790 */

#define LDH_E_NOT_INIT (-100)
#define LDH_E_ARG (-101)
#define LDH_E_HIP (-102)
#define LDH_E_NOMEM (-103)
#define LDH_E_CAPACITY (-104) /* caller's output slot too small */
#define LDH_E_TOO_LONG (-105) /* ldh_sync: a *_dev compress call met streams longer than its max_in_len */

/* Extra per-stream status beyond LibDeflate's codes: output slot too small (puff's 1). */
#define LDH_STATUS_OUTPUT_FULL 1

typedef struct ldh_dict ldh_dict;

/* ---- lifecycle ------------------------------------------------------------- */
/* One context (streams + workspace) on this HIP device; idempotent for the same device.  After ldh_init_devices()
 * it selects the pool's first context on `device` as the default one; a device that has no context yet while a
 * default exists gives LDH_E_ARG (device out of range: LDH_E_ARG; a failing HIP call: LDH_E_HIP). */
int ldh_init(int device);
void ldh_shutdown(void);
int ldh_device_count(void);
const char *ldh_last_error(void);
void *ldh_stream(void); /* hipStream_t the *_dev calls enqueue on (current context) */
/* Waits for the current context's stream.  LDH_E_TOO_LONG if a *_dev compress call since the last
 * ldh_sync met streams longer than its max_in_len: those have out_len 0 (they are never truncated).  The counter is
 * read and cleared on the library stream itself.  Callers that wait some other way (hipStreamSynchronize(ldh_stream()),
 * an event) never see this code: for them out_len[s] == 0 is the per-stream signal (no input compresses to 0 bytes). */
int ldh_sync(void);

/* ---- one library, many GPUs (host-side scatter, no inter-GPU collective) --------------
 * No reference counterpart (the reference is single-threaded, one string per call).
 * ldh_init_devices makes a pool of contexts, one per listed device (NULL / 0: every visible device; a
 * device listed twice gets two contexts).  From then on ldh_compress_batch and ldh_decompress_batch cut
 * their batch into contiguous, byte-balanced parts (ldh_partition_by_bytes), one per context, run them on
 * one host thread + three HIP streams per context, and results land in the caller's one buffer.  All other
 * calls (single strings, *_dev, profiling) use the context of ldh_init() -- the pool's first one if
 * ldh_init was not called -- or the pool member this thread chose with ldh_use_context(index) (-1: back to
 * the default).  Device pointers passed to *_dev calls must belong to the current context's device. */
int ldh_init_devices(const int *devices, int n);
int ldh_pool_size(void);
int ldh_use_context(int index);
/* first[0..parts]: part p takes streams [first[p], first[p+1]); pure host arithmetic. */
int ldh_partition_by_bytes(const uint64_t *in_off, size_t count, int parts, size_t *first);

/* ---- caller-pinned host memory (no reference counterpart) -------------------------------------------------
 * ldh_compress_batch / ldh_decompress_batch take ordinary pageable memory and stage it through pinned buffers of
 * their own (a gather before the copy in, a scatter after the copy out: host memcpy work that costs as much as the
 * PCIe transfers).  Inputs that lie inside a range pinned here are copied to the device straight from the caller's
 * memory, and results into output slots inside such a range straight from the device (rows as wide as the piece's
 * longest result: bytes of a slot beyond out_len[s], and slots of streams whose status is not 0, may then hold
 * anything).  ldh_host_register pins memory the caller owns (hipHostRegister); ldh_host_alloc returns pinned memory
 * (hipHostMalloc) to be released with ldh_host_free.  The small per-stream arrays need no pinning. */
int ldh_host_register(void *p, size_t bytes);
int ldh_host_unregister(void *p);
void *ldh_host_alloc(size_t bytes);
void ldh_host_free(void *p);

/* Largest output CompressDeflate/CompressZlib can produce for n input bytes at any
 * level/strategy; also the required out_stride granularity (multiple of 16). */
size_t ldh_compress_bound(size_t n);

/* ---- preset dictionary: LibDeflate:CreateDictionary (LD:493-596) -------------
 * The shim keeps the Lua-visible table {adler32,strlen,string_table,hash_tables};
 * this handle is the device-side copy of the same bytes (1..32768 of them). */
ldh_dict *ldh_dict_create(const uint8_t *bytes, size_t n);
void ldh_dict_free(ldh_dict *d);
uint32_t ldh_dict_adler32(const ldh_dict *d);
size_t ldh_dict_len(const ldh_dict *d);

/* ---- single-string forms (host pointers) -------------------------------------
 * Replaces CompressDeflateInternal (LD:1960-1966) / CompressZlibInternal
 * (LD:1970-2024): returns the stream and padding_bitlen. */
int ldh_compress(const uint8_t *in, size_t n, int fmt, int level, int strategy,
                 const ldh_dict *dict, uint8_t *out, size_t cap, size_t *out_len,
                 int *pad_bits);
/* Replaces DecompressDeflateInternal (LD:2648-2656) / DecompressZlibInternal
 * (LD:2660-2727).  *status 0: out_len bytes written, *bytes_left = unprocessed input
 * bytes; otherwise the LibDeflate status (2, -1..-17) or LDH_STATUS_OUTPUT_FULL. */
int ldh_decompress(const uint8_t *in, size_t n, int fmt, const ldh_dict *dict,
                   uint8_t *out, size_t cap, size_t *out_len, size_t *bytes_left,
                   int *status);
/* LibDeflate:Adler32 (LD:401-442) */
int ldh_adler32(const uint8_t *s, size_t n, uint32_t *adler);
/* LibDeflate:EncodeForPrint (LD:3311-3353); out needs (8n+5)/6 bytes. */
int ldh_print6_encode(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                      size_t *out_len);
/* LibDeflate:DecodeForPrint (LD:3365-3417); *ok = 0 stands for the nil return. */
int ldh_print6_decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                      size_t *out_len, int *ok);

/* ---- batched forms (the "batched entry point" of the north star) --------------
 * Stream s is in_base[in_off[s] .. in_off[s+1]).  Its result goes to
 * out_base + s*out_stride (out_stride % 16 == 0), length out_len[s].
 * Host-pointer forms copy in/out and synchronise; *_dev forms take DEVICE pointers for
 * every array argument and only enqueue.  level must be 0..9 for *_dev (host forms also
 * accept LDH_LEVEL_DEFAULT and resolve it per stream). */
int ldh_compress_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt,
                       int level, int strategy, const ldh_dict *dict, uint8_t *out_base,
                       size_t out_stride, uint32_t *out_len, uint8_t *pad_bits);
int ldh_compress_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                           size_t max_in_len, int fmt, int level, int strategy,
                           const ldh_dict *dict, uint8_t *d_out, size_t out_stride,
                           uint32_t *d_out_len, uint8_t *d_pad_bits);

int ldh_decompress_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt,
                         const ldh_dict *dict, uint8_t *out_base, size_t out_stride,
                         uint32_t *out_len, uint32_t *bytes_left, int32_t *status);
int ldh_decompress_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                             int fmt, const ldh_dict *dict, uint8_t *d_out, size_t out_stride,
                             uint32_t *d_out_len, uint32_t *d_bytes_left, int32_t *d_status);

/* ldh_decompress_batch_dev for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc,
 * code that runs beside a HIP graph capture).  No counterpart in the reference.  Same arguments and results;
 * it calls no hipMalloc / hipFree: the per-stream scratch of the zlib trailer check lives in static device
 * memory (8 bytes x LDH_NOALLOC_MAX_STREAMS per GPU, held until the process ends).  count above
 * LDH_NOALLOC_MAX_STREAMS gives LDH_E_CAPACITY; a dictionary must already be resident on the device
 * (it is after ldh_dict_create on this context or any earlier call with it), else LDH_E_ARG.  Profiling is
 * off for the call.  The work is enqueued on the library's own stream (ldh_stream()), like every *_dev call: it is
 * NOT captured into a graph the caller is recording on another stream; and the first launch of a kernel on a device
 * may still load its code object, so run one call of each format once at start-up.  Calls on one GPU share the
 * scratch: calls made through one context are ordered by its stream; a call made through another context of the
 * same GPU (ldh_init_devices with a device listed twice) is made to wait for the previous one by an event. */
#define LDH_NOALLOC_MAX_STREAMS 65536
int ldh_decompress_batch_dev_noalloc(const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                                     int fmt, const ldh_dict *dict, uint8_t *d_out, size_t out_stride,
                                     uint32_t *d_out_len, uint32_t *d_bytes_left, int32_t *d_status);

/* Same, for inputs laid out as ldh_compress_batch_dev leaves them: stream s is
 * d_in + s*in_stride, d_in_len[s] bytes (so a device-resident round trip needs no repacking). */
int ldh_decompress_slots_dev(const uint8_t *d_in, size_t in_stride, const uint32_t *d_in_len,
                             size_t count, int fmt, const ldh_dict *dict, uint8_t *d_out,
                             size_t out_stride, uint32_t *d_out_len, uint32_t *d_bytes_left,
                             int32_t *d_status);

int ldh_adler32_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count,
                      uint32_t *adler);
int ldh_adler32_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                          uint32_t *d_adler);

int ldh_print6_encode_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count,
                            uint8_t *out_base, size_t out_stride, uint32_t *out_len);
int ldh_print6_encode_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                                uint8_t *d_out, size_t out_stride, uint32_t *d_out_len);
/* ok[s] = 0 stands for DecodeForPrint's nil. */
int ldh_print6_decode_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count,
                            uint8_t *out_base, size_t out_stride, uint32_t *out_len,
                            uint8_t *ok);
int ldh_print6_decode_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t count,
                                uint8_t *d_out, size_t out_stride, uint32_t *d_out_len,
                                uint8_t *d_ok);

/* ---- CreateCodec and the WoW channel codecs -----------------------
 * LibDeflate:CreateCodec(reserved_chars, escape_chars, map_chars) (LD:2921-3056).  Returns NULL
 * and sets *err to the reference's second return value ("No escape characters supplied." ...) when
 * the codec cannot be created; the type check and its usage error stay in the shim.
 * EncodeForWoWAddonChannel / ChatChannel (LD:3058-3158) are this with the reference's fixed
 * arguments ("\000","\001","" and LD:3118-3124). */
typedef struct ldh_codec ldh_codec;
ldh_codec *ldh_codec_create(const uint8_t *reserved, size_t nr, const uint8_t *escape, size_t ne,
                            const uint8_t *map, size_t nm, const char **err);
void ldh_codec_free(ldh_codec *c);
/* codec:Encode (LD:3030-3037); out_stride >= 2 * longest stream (and % 16 == 0). */
int ldh_codec_encode_batch(const ldh_codec *c, const uint8_t *in_base, const uint64_t *in_off,
                           size_t count, uint8_t *out_base, size_t out_stride, uint32_t *out_len);
int ldh_codec_encode_batch_dev(const ldh_codec *c, const uint8_t *d_in, const uint64_t *d_in_off,
                               size_t count, uint8_t *d_out, size_t out_stride, uint32_t *d_out_len);
/* codec:Decode (LD:3042-3053); ok[s] = 0 stands for nil; out_stride >= longest stream. */
int ldh_codec_decode_batch(const ldh_codec *c, const uint8_t *in_base, const uint64_t *in_off,
                           size_t count, uint8_t *out_base, size_t out_stride, uint32_t *out_len,
                           uint8_t *ok);
int ldh_codec_decode_batch_dev(const ldh_codec *c, const uint8_t *d_in, const uint64_t *d_in_off,
                               size_t count, uint8_t *d_out, size_t out_stride, uint32_t *d_out_len,
                               uint8_t *d_ok);

/* ---- measurement support (bench.py) -------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the library
 * stream; ldh_profile_get returns the accumulated milliseconds and launch count of one
 * kernel by name since the last ldh_profile_reset.  No reference counterpart. */
int ldh_profile_enable(int on);
int ldh_profile_reset(void);
int ldh_profile_count(void);
int ldh_profile_get(int index, char *name, size_t name_cap, double *ms, uint64_t *launches);
/* hipEvent pair on the library stream for whole-pipeline timing */
int ldh_timer_start(void);
int ldh_timer_stop(double *ms);

/* ---- test / inspection hooks (no reference counterpart; used by tests/ only) --
 * Copies intermediate products of the last ldh_compress_batch* call for stream s to
 * host memory.  what: 0 prev-link deltas (uint16 per virtual position), 1 per-position
 * best match (uint32: len<<16|dist), 2 tokens (uint32), 3 block table (opaque bytes). */
int ldh_debug_fetch(int what, size_t stream, void *dst, size_t cap_bytes, size_t *got_bytes);
/* Device, pinned-host and event allocations the library has made since it was loaded (hipMalloc, hipHostMalloc,
 * hipEventCreate calls): a test can show that an entry point allocated nothing by reading it before and after. */
uint64_t ldh_debug_alloc_count(void);

#ifdef __cplusplus
}
#endif
#endif
