--[[ LibDeflate-compatible shim over libdeflate_hip.so (LuaJIT 2.1 FFI).

Drop-in for require("LibDeflate") on the hot path only: CompressDeflate / CompressZlib /
Compress*WithDict / DecompressDeflate / DecompressZlib / Decompress*WithDict / Adler32 /
CreateDictionary / EncodeForPrint / DecodeForPrint / CreateCodec / the WoW channel codecs.  Argument validation and the usage-error
strings are kept in Lua exactly as the reference words them; everything after validation goes
through the C ABI of include/libdeflate_hip.h.  The batched entry point is a module of its own,
require("LibDeflateBatch") (lua/LibDeflateBatch.lua): the reference's TestExported demands exactly its
22 keys on this table, and `require` hands back one value only, so it can be neither a 23rd key nor a
second return value.

NOT TESTED in the build image: no Lua/LuaJIT interpreter exists there (SURVEY.md 8.c).
]]
local ffi = require("ffi")
ffi.cdef [[
typedef struct ldh_dict ldh_dict;
int ldh_init(int device);
const char *ldh_last_error(void);
size_t ldh_compress_bound(size_t n);
ldh_dict *ldh_dict_create(const uint8_t *bytes, size_t n);
void ldh_dict_free(ldh_dict *d);
int ldh_compress(const uint8_t *in, size_t n, int fmt, int level, int strategy, const ldh_dict *dict,
                 uint8_t *out, size_t cap, size_t *out_len, int *pad_bits);
int ldh_decompress(const uint8_t *in, size_t n, int fmt, const ldh_dict *dict, uint8_t *out, size_t cap,
                   size_t *out_len, size_t *bytes_left, int *status);
int ldh_adler32(const uint8_t *s, size_t n, uint32_t *adler);
int ldh_print6_encode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);
int ldh_print6_decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, int *ok);
typedef struct ldh_codec ldh_codec;
ldh_codec *ldh_codec_create(const uint8_t *reserved, size_t nr, const uint8_t *escape, size_t ne,
                            const uint8_t *map, size_t nm, const char **err);
void ldh_codec_free(ldh_codec *c);
int ldh_codec_encode_batch(const ldh_codec *c, const uint8_t *in_base, const uint64_t *in_off, size_t count,
                           uint8_t *out_base, size_t out_stride, uint32_t *out_len);
int ldh_codec_decode_batch(const ldh_codec *c, const uint8_t *in_base, const uint64_t *in_off, size_t count,
                           uint8_t *out_base, size_t out_stride, uint32_t *out_len, uint8_t *ok);
int ldh_compress_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt, int level,
                       int strategy, const ldh_dict *dict, uint8_t *out_base, size_t out_stride,
                       uint32_t *out_len, uint8_t *pad_bits);
int ldh_decompress_batch(const uint8_t *in_base, const uint64_t *in_off, size_t count, int fmt,
                         const ldh_dict *dict, uint8_t *out_base, size_t out_stride, uint32_t *out_len,
                         uint32_t *bytes_left, int32_t *status);
]]
local C = ffi.load(os.getenv("LIBDEFLATE_HIP_SO") or "libdeflate_hip.so")
local function check(rc)
  if rc ~= 0 then error("libdeflate_hip: " .. ffi.string(C.ldh_last_error()), 3) end
end
check(C.ldh_init(tonumber(os.getenv("LIBDEFLATE_HIP_DEVICE") or "0")))

local LibDeflate = {_VERSION = "1.0.2-release", _MAJOR = "LibDeflate", _MINOR = 3,
                    _COPYRIGHT = "LibDeflate 1.0.2-release Copyright (C) 2018-2021 Haoqian He. zlib License"}
local STRATEGY = {fixed = 1, huffman_only = 2, dynamic = 3}
local LEVEL_OK = {[0]=true,true,true,true,true,true,true,true,true,true}
local dict_handles = setmetatable({}, {__mode = "k"})  -- dictionary table -> device handle

local function IsValidDictionary(d)  -- LibDeflate.lua:602-615
  if type(d) ~= "table" then
    return false, ("'dictionary' - table expected got '%s'."):format(type(d))
  end
  if type(d.adler32) ~= "number" or type(d.string_table) ~= "table" or type(d.strlen) ~= "number"
    or d.strlen <= 0 or d.strlen > 32768 or d.strlen ~= #d.string_table or type(d.hash_tables) ~= "table" then
    return false, ("'dictionary' - corrupted dictionary."):format(type(d))
  end
  return true, ""
end
local function IsValidArguments(str, check_dictionary, dictionary, check_configs, configs)  -- :679-714
  if type(str) ~= "string" then return false, ("'str' - string expected got '%s'."):format(type(str)) end
  if check_dictionary then
    local ok, err = IsValidDictionary(dictionary)
    if not ok then return false, err end
  end
  if check_configs then
    local t = type(configs)
    if t ~= "nil" and t ~= "table" then
      return false, ("'configs' - nil or table expected got '%s'."):format(type(configs))
    end
    if t == "table" then
      for k, v in pairs(configs) do
        if k ~= "level" and k ~= "strategy" then
          return false, ("'configs' - unsupported table key in the configs: '%s'."):format(k)
        elseif k == "level" and not LEVEL_OK[v] then
          return false, ("'configs' - unsupported 'level': %s."):format(tostring(v))
        elseif k == "strategy" and v ~= "fixed" and v ~= "huffman_only" and v ~= "dynamic" then
          return false, ("'configs' - unsupported 'strategy': '%s'."):format(tostring(v))
        end
      end
    end
  end
  return true, ""
end

local function handle_of(dictionary)
  if not dictionary then return nil end
  local h = dict_handles[dictionary]
  if not h then
    local n = dictionary.strlen
    local buf = ffi.new("uint8_t[?]", n)
    for i = 1, n do buf[i - 1] = dictionary.string_table[i] end
    h = C.ldh_dict_create(buf, n)
    if h == nil then error("libdeflate_hip: " .. ffi.string(C.ldh_last_error()), 3) end
    h = ffi.gc(h, C.ldh_dict_free)
    dict_handles[dictionary] = h
  end
  return h
end

function LibDeflate:Adler32(str)
  if type(str) ~= "string" then
    error(("Usage: LibDeflate:Adler32(str): 'str' - string expected got '%s'."):format(type(str)), 2)
  end
  local out = ffi.new("uint32_t[1]")
  check(C.ldh_adler32(str, #str, out))
  return tonumber(out[0])
end

function LibDeflate:CreateDictionary(str, strlen, adler32)  -- same table shape as LibDeflate.lua:531-595
  local pre = "Usage: LibDeflate:CreateDictionary(str, strlen, adler32):"
  if type(str) ~= "string" then error((pre .. " 'str' - string expected got '%s'."):format(type(str)), 2) end
  if type(strlen) ~= "number" then error((pre .. " 'strlen' - number expected got '%s'."):format(type(strlen)), 2) end
  if type(adler32) ~= "number" then error((pre .. " 'adler32' - number expected got '%s'."):format(type(adler32)), 2) end
  if strlen ~= #str then
    error((pre .. " 'strlen' does not match the actual length of 'str'. 'strlen': %u, '#str': %u ." ..
           " Please check if 'str' is modified unintentionally."):format(strlen, #str))
  end
  if strlen == 0 then error(pre .. " 'str' - Empty string is not allowed.", 2) end
  if strlen > 32768 then
    error((pre .. " 'str' - string longer than 32768 bytes is not allowed. Got %d bytes."):format(strlen), 2)
  end
  local actual = self:Adler32(str)
  if adler32 % 4294967296 ~= actual % 4294967296 then
    error((pre .. " 'adler32' does not match the actual adler32 of 'str'. 'adler32': %u, 'Adler32(str)': %u ." ..
           " Please check if 'str' is modified unintentionally."):format(adler32, actual))
  end
  local d = {adler32 = adler32, strlen = strlen, string_table = {}, hash_tables = {}}
  for i = 1, strlen do d.string_table[i] = str:byte(i) end
  for i = 1, strlen - 2 do  -- kept so that callers / tests that inspect hash_tables keep working
    local h = d.string_table[i] * 65536 + d.string_table[i + 1] * 256 + d.string_table[i + 2]
    local t = d.hash_tables[h]
    if not t then t = {}; d.hash_tables[h] = t end
    t[#t + 1] = i - strlen
  end
  return d
end

local function compress(str, fmt, dictionary, configs)
  local level, strategy = -1, 0
  if configs then
    if configs.level then level = configs.level end
    if configs.strategy then strategy = STRATEGY[configs.strategy] end
  end
  local cap = tonumber(C.ldh_compress_bound(#str))
  local out = ffi.new("uint8_t[?]", cap)
  local ol, pb = ffi.new("size_t[1]"), ffi.new("int[1]")
  check(C.ldh_compress(str, #str, fmt, level, strategy, handle_of(dictionary), out, cap, ol, pb))
  return ffi.string(out, ol[0]), pb[0]
end
local function decompress(str, fmt, dictionary)
  local cap = math.max(64, 4 * #str + 64)
  local h = handle_of(dictionary)
  while true do
    local out = ffi.new("uint8_t[?]", cap)
    local ol, bl, st = ffi.new("size_t[1]"), ffi.new("size_t[1]"), ffi.new("int[1]")
    check(C.ldh_decompress(str, #str, fmt, h, out, cap, ol, bl, st))
    if st[0] == 1 then  -- LDH_STATUS_OUTPUT_FULL: Lua strings simply grow -- up to the library's limit
      if cap >= 0xFFF00000 then error("libdeflate_hip: output of 4 GiB or more is not supported", 3) end
      cap = math.min(cap * 4, 0xFFF00000)
    elseif st[0] ~= 0 then return nil, st[0]
    else return ffi.string(out, ol[0]), tonumber(bl[0]) end
  end
end

local function wrap(name, sig, fmt, with_dict, is_compress)
  LibDeflate[name] = function(self, str, a, b)
    local dictionary, configs
    if with_dict then dictionary, configs = a, b else configs = a end
    local ok, err = IsValidArguments(str, with_dict, dictionary, is_compress, configs)
    if not ok then error(("Usage: LibDeflate:%s(%s): "):format(name, sig) .. err, 2) end
    if is_compress then return compress(str, fmt, dictionary, configs) end
    return decompress(str, fmt, dictionary)
  end
end
wrap("CompressDeflate", "str, configs", 0, false, true)
wrap("CompressDeflateWithDict", "str, dictionary, configs", 0, true, true)
wrap("CompressZlib", "str, configs", 1, false, true)
wrap("CompressZlibWithDict", "str, dictionary, configs", 1, true, true)
wrap("DecompressDeflate", "str", 0, false, false)
wrap("DecompressDeflateWithDict", "str, dictionary", 0, true, false)
wrap("DecompressZlib", "str", 1, false, false)
wrap("DecompressZlibWithDict", "str, dictionary", 1, true, false)

function LibDeflate:EncodeForPrint(str)
  if type(str) ~= "string" then
    error(("Usage: LibDeflate:EncodeForPrint(str): 'str' - string expected got '%s'."):format(type(str)), 2)
  end
  local cap = math.floor((#str * 8 + 5) / 6) + 16
  local out, ol = ffi.new("uint8_t[?]", cap), ffi.new("size_t[1]")
  check(C.ldh_print6_encode(str, #str, out, cap, ol))
  return ffi.string(out, ol[0])
end
function LibDeflate:DecodeForPrint(str)
  if type(str) ~= "string" then
    error(("Usage: LibDeflate:DecodeForPrint(str): 'str' - string expected got '%s'."):format(type(str)), 2)
  end
  local cap = #str + 16
  local out, ol, ok = ffi.new("uint8_t[?]", cap), ffi.new("size_t[1]"), ffi.new("int[1]")
  check(C.ldh_print6_decode(str, #str, out, cap, ol, ok))
  if ok[0] == 0 then return nil end
  return ffi.string(out, ol[0])
end

-- CreateCodec and the WoW channel codecs (reference :2921-3158): tables live in the library
function LibDeflate:CreateCodec(reserved_chars, escape_chars, map_chars)
  if type(reserved_chars) ~= "string" or type(escape_chars) ~= "string" or type(map_chars) ~= "string" then
    error("Usage: LibDeflate:CreateCodec(reserved_chars," .. " escape_chars, map_chars):" ..
            " All arguments must be string.", 2)
  end
  local err = ffi.new("const char *[1]")
  local h = C.ldh_codec_create(reserved_chars, #reserved_chars, escape_chars, #escape_chars, map_chars, #map_chars, err)
  if h == nil then return nil, ffi.string(err[0]) end
  h = ffi.gc(h, C.ldh_codec_free)
  local codec = {}
  local function run(str, decode)
    local off = ffi.new("uint64_t[2]", 0, #str)
    local stride = math.floor(((decode and #str or 2 * #str) + 31) / 16) * 16
    local out, ol, ok = ffi.new("uint8_t[?]", stride), ffi.new("uint32_t[1]"), ffi.new("uint8_t[1]")
    if decode then
      check(C.ldh_codec_decode_batch(h, str, off, 1, out, stride, ol, ok))
      if ok[0] == 0 then return nil end
    else
      check(C.ldh_codec_encode_batch(h, str, off, 1, out, stride, ol))
    end
    return ffi.string(out, ol[0])
  end
  function codec:Encode(str)
    if type(str) ~= "string" then
      error(("Usage: codec:Encode(str):" .. " 'str' - string expected got '%s'."):format(type(str)), 2)
    end
    return run(str, false)
  end
  function codec:Decode(str)
    if type(str) ~= "string" then
      error(("Usage: codec:Decode(str):" .. " 'str' - string expected got '%s'."):format(type(str)), 2)
    end
    return run(str, true)
  end
  return codec
end
local _addon_codec, _chat_codec
local function addon_codec()
  _addon_codec = _addon_codec or LibDeflate:CreateCodec("\000", "\001", "")
  return _addon_codec
end
local function chat_codec()
  if not _chat_codec then
    local r = {}
    for i = 128, 255 do r[#r + 1] = string.char(i) end
    _chat_codec = LibDeflate:CreateCodec("sS\000\010\013\124%" .. table.concat(r), "\029\031", "\015\020")
  end
  return _chat_codec
end
for name, spec in pairs({EncodeForWoWAddonChannel = {addon_codec, "Encode"}, DecodeForWoWAddonChannel = {addon_codec, "Decode"},
                         EncodeForWoWChatChannel = {chat_codec, "Encode"}, DecodeForWoWChatChannel = {chat_codec, "Decode"}}) do
  LibDeflate[name] = function(_, str)
    if type(str) ~= "string" then
      error(("Usage: LibDeflate:" .. name .. "(str):" .. " 'str' - string expected got '%s'."):format(type(str)), 2)
    end
    local codec = spec[1]()
    return codec[spec[2]](codec, str)
  end
end

-- For test (reference :3419-3433): the same six names.
local function LoadStringToTable(str, t, start, stop, offset)  -- :1184-1197: str[i] -> t[i - offset], start <= i <= stop
  for i = start, stop do t[i - offset] = str:byte(i) end
  return t
end
local function IsEqualAdler32(actual, expected)  -- :447-449
  return (actual % 4294967296) == (expected % 4294967296)
end
local _byte_to_6bit_char, _6bit_to_byte = {}, {}  -- :3165-3297: a-z A-Z 0-9 ( )
do
  local alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789()"
  for i = 0, 63 do
    _byte_to_6bit_char[i] = alphabet:sub(i + 1, i + 1)
    _6bit_to_byte[alphabet:byte(i + 1)] = i
  end
end
local function InternalClearCache()
  _chat_codec = nil
  _addon_codec = nil
end
LibDeflate.internals = {
  LoadStringToTable = LoadStringToTable,
  IsValidDictionary = IsValidDictionary,
  IsEqualAdler32 = IsEqualAdler32,
  _byte_to_6bit_char = _byte_to_6bit_char,
  _6bit_to_byte = _6bit_to_byte,
  InternalClearCache = InternalClearCache,
}

-- What lua/LibDeflateBatch.lua builds on (not a key of the method table: kept in the loader's registry)
package.loaded["LibDeflate.__hip"] = {C = C, ffi = ffi, check = check, handle_of = handle_of, STRATEGY = STRATEGY,
                                      IsValidArguments = IsValidArguments}
return LibDeflate
