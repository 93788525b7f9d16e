--[[ Batched entry point of the libdeflate_hip.so shim (the north star's "plus a batched entry point"):

  local LibDeflate = require("LibDeflate")       -- lua/LibDeflate.lua, the 22-key drop-in
  local Batch = require("LibDeflateBatch")

  local outs, pads   = Batch.Compress(list, fmt, configs, dictionary)   -- fmt 0 raw deflate / 1 zlib
  local outs, second = Batch.Decompress(list, fmt, dictionary)          -- outs[i] string or nil; second[i] bytes left or status
  local sums         = Batch.Adler32(list)
  local texts        = Batch.EncodeForPrint(list)
  local datas        = Batch.DecodeForPrint(list)                       -- datas[i] string or nil

  -- caller-owned memory (cdata pointers: malloc'ed, mmap'ed or pinned buffers; nothing goes through Lua strings):
  local rc = Batch.CompressBuffers(in_ptr, off_ptr, n, fmt, configs, dictionary, out_ptr, out_stride, len_ptr, pad_ptr)
  local rc = Batch.DecompressBuffers(in_ptr, off_ptr, n, fmt, dictionary, out_ptr, out_stride, len_ptr, left_ptr, status_ptr)
  local buf  = Batch.Alloc(nbytes)               -- malloc'ed uint8_t*, freed by the garbage collector (ffi.gc)
  local buf  = Batch.AllocPinned(nbytes)         -- page-locked (ldh_host_alloc): the *Buffers calls copy from / to it directly
  local need = Batch.CompressBound(maxlen)       -- out_stride for CompressBuffers

The list forms build ONE Lua string (table.concat) and ONE ffi.new array for the results: both live on LuaJIT's
garbage-collected heap, which a non-GC64 build limits to 1-2 GiB in all.  Use them up to a few hundred MiB per call
(inputs + count * out_stride); beyond that -- the bench's batch is 1 GiB in and 1.3 GiB of slots -- use the
*Buffers forms, which hand the caller's pointers straight to ldh_compress_batch / ldh_decompress_batch.

Each call is ONE trip through the C ABI (ldh_*_batch of include/libdeflate_hip.h): one copy in, one set of
kernel launches over all strings, one copy out -- and, after ldh_init_devices, split over every GPU of the
node by the library.  The reference has no counterpart (one string per call); argument checks follow
its IsValidArguments per element.  Shares the FFI handle and the dictionary cache of lua/LibDeflate.lua.

NOT TESTED in the build image: no Lua/LuaJIT interpreter exists there (SURVEY.md 8.c); tests/test_lua_shim.py
checks its ffi.cdef against the header token for token.
]]
require("LibDeflate")
local hip = package.loaded["LibDeflate.__hip"]
local ffi, C, check, handle_of = hip.ffi, hip.C, hip.check, hip.handle_of
ffi.cdef [[
int ldh_init_devices(const int *devices, int n);
int ldh_pool_size(void);
int ldh_adler32_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, uint32_t *adler);
int ldh_print6_encode_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, uint8_t *out_base,
                            size_t out_stride, uint32_t *out_len);
int ldh_print6_decode_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, uint8_t *out_base,
                            size_t out_stride, uint32_t *out_len, uint8_t *ok);
]]
ffi.cdef [[
void *malloc(size_t size);
void free(void *ptr);
]]
ffi.cdef [[
void *ldh_host_alloc(size_t bytes);
void ldh_host_free(void *p);
]]

local Batch = {}

-- strings back to back + offsets (count + 1) + the longest length
local function pack(strings, fname)
  if type(strings) ~= "table" then
    error(("Usage: LibDeflateBatch.%s(strings, ...): 'strings' - table expected got '%s'."):format(fname, type(strings)), 3)
  end
  local n = #strings
  local off = ffi.new("uint64_t[?]", n + 1)
  local maxlen = 0
  for i = 1, n do
    local s = strings[i]
    if type(s) ~= "string" then
      error(("Usage: LibDeflateBatch.%s(strings, ...): 'strings[%d]' - string expected got '%s'."):format(fname, i, type(s)), 3)
    end
    off[i] = off[i - 1] + #s
    if #s > maxlen then maxlen = #s end
  end
  return table.concat(strings), off, n, maxlen
end
local function round16(x) return math.floor((x + 15) / 16) * 16 end

-- Use every GPU of the node for the calls below (optional; default is the one device of LIBDEFLATE_HIP_DEVICE).
function Batch.UseAllDevices()
  if C.ldh_pool_size() == 0 then check(C.ldh_init_devices(nil, 0)) end
  return C.ldh_pool_size()
end

function Batch.Compress(strings, fmt, configs, dictionary)
  local ok, err = hip.IsValidArguments("", dictionary ~= nil, dictionary, true, configs)
  if not ok then error("Usage: LibDeflateBatch.Compress(strings, fmt, configs, dictionary): " .. err, 2) end
  local data, off, n, maxlen = pack(strings, "Compress")
  local level, strategy = -1, 0
  if configs then level = configs.level or -1; strategy = hip.STRATEGY[configs.strategy] or 0 end
  local res, pads = {}, {}
  if n == 0 then return res, pads end
  local stride = tonumber(C.ldh_compress_bound(maxlen))
  local out = ffi.new("uint8_t[?]", n * stride)
  local ol, pb = ffi.new("uint32_t[?]", n), ffi.new("uint8_t[?]", n)
  check(C.ldh_compress_batch(data, off, n, fmt or 0, level, strategy, handle_of(dictionary), out, stride, ol, pb))
  for i = 1, n do res[i] = ffi.string(out + (i - 1) * stride, ol[i - 1]); pads[i] = pb[i - 1] end
  return res, pads
end

-- ---- caller-owned memory ------------------------------------------------------------------------------------
-- The same two calls over cdata pointers: stream i is in_ptr[off_ptr[i] .. off_ptr[i+1]) (off_ptr: uint64_t[n+1]),
-- its result goes to out_ptr + i*out_stride with its length in len_ptr[i] (uint32_t[n]).  Nothing is copied through
-- Lua strings and nothing is allocated on the Lua heap, so a batch may be as large as host memory allows.
-- Returns the number of streams; results are read from the caller's buffers.
local function is_ptr(p) return type(p) == "cdata" end
local function need_ptrs(fname, ...)
  local names = {...}
  for i = 1, #names, 2 do
    if not is_ptr(names[i + 1]) then
      error(("Usage: LibDeflateBatch.%s: '%s' - cdata pointer expected got '%s'."):format(fname, names[i], type(names[i + 1])), 3)
    end
  end
end

function Batch.CompressBound(maxlen) return tonumber(C.ldh_compress_bound(maxlen)) end

-- nbytes of malloc'ed memory as uint8_t* (outside LuaJIT's heap), released when the cdata is collected
function Batch.Alloc(nbytes)
  local p = ffi.C.malloc(nbytes)
  if p == nil then error("LibDeflateBatch.Alloc: out of memory (" .. tostring(nbytes) .. " bytes)", 2) end
  return ffi.gc(ffi.cast("uint8_t *", p), ffi.C.free)
end

-- nbytes of page-locked host memory (ldh_host_alloc) as uint8_t*: the library copies between such a buffer and the GPU
-- directly, without staging it through pinned buffers of its own (about a fifth faster at 2048 x 128 KiB)
function Batch.AllocPinned(nbytes)
  local p = C.ldh_host_alloc(nbytes)
  if p == nil then error("LibDeflateBatch.AllocPinned: cannot pin " .. tostring(nbytes) .. " bytes", 2) end
  return ffi.gc(ffi.cast("uint8_t *", p), C.ldh_host_free)
end

-- out_stride: a multiple of 16, at least Batch.CompressBound(longest stream); pad_ptr: uint8_t[n]
function Batch.CompressBuffers(in_ptr, off_ptr, n, fmt, configs, dictionary, out_ptr, out_stride, len_ptr, pad_ptr)
  local ok, err = hip.IsValidArguments("", dictionary ~= nil, dictionary, true, configs)
  if not ok then error("Usage: LibDeflateBatch.CompressBuffers(in_ptr, off_ptr, n, fmt, configs, dictionary, out_ptr, out_stride, len_ptr, pad_ptr): " .. err, 2) end
  need_ptrs("CompressBuffers", "in_ptr", in_ptr, "off_ptr", off_ptr, "out_ptr", out_ptr, "len_ptr", len_ptr, "pad_ptr", pad_ptr)
  local level, strategy = -1, 0
  if configs then level = configs.level or -1; strategy = hip.STRATEGY[configs.strategy] or 0 end
  if n == 0 then return 0 end
  check(C.ldh_compress_batch(ffi.cast("const uint8_t *", in_ptr), ffi.cast("const uint64_t *", off_ptr), n, fmt or 0, level,
                             strategy, handle_of(dictionary), ffi.cast("uint8_t *", out_ptr), out_stride,
                             ffi.cast("uint32_t *", len_ptr), ffi.cast("uint8_t *", pad_ptr)))
  return n
end

-- left_ptr: uint32_t[n] (unprocessed input bytes where status is 0), status_ptr: int32_t[n] (0, LibDeflate's codes
-- 2 / -1..-17, or 1 = the slot of out_stride bytes was too small)
function Batch.DecompressBuffers(in_ptr, off_ptr, n, fmt, dictionary, out_ptr, out_stride, len_ptr, left_ptr, status_ptr)
  if dictionary ~= nil then
    local ok, err = hip.IsValidArguments("", true, dictionary)
    if not ok then error("Usage: LibDeflateBatch.DecompressBuffers(in_ptr, off_ptr, n, fmt, dictionary, out_ptr, out_stride, len_ptr, left_ptr, status_ptr): " .. err, 2) end
  end
  need_ptrs("DecompressBuffers", "in_ptr", in_ptr, "off_ptr", off_ptr, "out_ptr", out_ptr, "len_ptr", len_ptr, "left_ptr", left_ptr,
            "status_ptr", status_ptr)
  if n == 0 then return 0 end
  check(C.ldh_decompress_batch(ffi.cast("const uint8_t *", in_ptr), ffi.cast("const uint64_t *", off_ptr), n, fmt or 0,
                               handle_of(dictionary), ffi.cast("uint8_t *", out_ptr), out_stride, ffi.cast("uint32_t *", len_ptr),
                               ffi.cast("uint32_t *", left_ptr), ffi.cast("int32_t *", status_ptr)))
  return n
end

-- max_out (optional): a hard limit on one result's size; without it slots grow (x4) until every stream fits,
-- up to the library's 4 GiB limit.
function Batch.Decompress(strings, fmt, dictionary, max_out)
  if dictionary ~= nil then
    local ok, err = hip.IsValidArguments("", true, dictionary)
    if not ok then error("Usage: LibDeflateBatch.Decompress(strings, fmt, dictionary, max_out): " .. err, 2) end
  end
  local data, off, n, maxlen = pack(strings, "Decompress")
  local res, second = {}, {}
  if n == 0 then return res, second end
  local h = handle_of(dictionary)
  local cap = max_out or math.max(64, 8 * maxlen + 64)
  local todo = {}
  for i = 1, n do todo[i] = i end
  while #todo > 0 do
    local m, stride = #todo, round16(cap)
    local d, o = data, off
    if m < n then  -- only the streams whose slot was too small go round again
      local sub = {}
      for k = 1, m do sub[k] = strings[todo[k]] end
      d, o = pack(sub, "Decompress")
    end
    local out = ffi.new("uint8_t[?]", m * stride)
    local ol, bl, st = ffi.new("uint32_t[?]", m), ffi.new("uint32_t[?]", m), ffi.new("int32_t[?]", m)
    check(C.ldh_decompress_batch(d, o, m, fmt or 0, h, out, stride, ol, bl, st))
    local again = {}
    for k = 1, m do
      local i, s = todo[k], st[k - 1]
      if s == 1 and not max_out and cap < 0xFFF00000 then again[#again + 1] = i   -- LDH_STATUS_OUTPUT_FULL
      elseif s ~= 0 then res[i] = nil; second[i] = s
      else res[i] = ffi.string(out + (k - 1) * stride, ol[k - 1]); second[i] = bl[k - 1] end
    end
    todo = again
    cap = math.min(cap * 4, 0xFFF00000)
  end
  return res, second
end

function Batch.Adler32(strings)
  local data, off, n = pack(strings, "Adler32")
  local res = {}
  if n == 0 then return res end
  local out = ffi.new("uint32_t[?]", n)
  check(C.ldh_adler32_batch(data, off, n, out))
  for i = 1, n do res[i] = tonumber(out[i - 1]) end
  return res
end

function Batch.EncodeForPrint(strings)
  local data, off, n, maxlen = pack(strings, "EncodeForPrint")
  local res = {}
  if n == 0 then return res end
  local stride = round16(math.floor((maxlen * 8 + 5) / 6) + 16)
  local out, ol = ffi.new("uint8_t[?]", n * stride), ffi.new("uint32_t[?]", n)
  check(C.ldh_print6_encode_batch(data, off, n, out, stride, ol))
  for i = 1, n do res[i] = ffi.string(out + (i - 1) * stride, ol[i - 1]) end
  return res
end

function Batch.DecodeForPrint(strings)
  local data, off, n, maxlen = pack(strings, "DecodeForPrint")
  local res = {}
  if n == 0 then return res end
  local stride = round16(maxlen + 16)
  local out, ol, ok = ffi.new("uint8_t[?]", n * stride), ffi.new("uint32_t[?]", n), ffi.new("uint8_t[?]", n)
  check(C.ldh_print6_decode_batch(data, off, n, out, stride, ol, ok))
  for i = 1, n do
    if ok[i - 1] ~= 0 then res[i] = ffi.string(out + (i - 1) * stride, ol[i - 1]) end  -- else nil, as the reference returns
  end
  return res
end

return Batch
