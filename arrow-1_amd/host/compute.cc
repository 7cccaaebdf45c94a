// compute.cc -- runtime of the C++ host mirror: Status, data model glue, ExecBatchIterator,
// Function dispatch, executors, FunctionRegistry, CallFunction.  Follows the control flow of the
// reference's compute/{exec,function,registry}.cc (cited per function); no kernel lives here.
#include "amdarrow_compute.h"

#include <algorithm>
#include <cstring>
#include <limits>
#include <sstream>
#include <vector>

namespace amdarrow {

// ---- Status -------------------------------------------------------------------------------------
Status Status::FromAbi(int code) {
  if (code == AMDARROW_OK) return Status::OK();
  return Status(static_cast<StatusCode>(code), amdarrow_last_error());
}
std::string Status::ToString() const {
  static const char* names[] = {"OK", "Out of memory", "Key error", "Type error", "Invalid", "IOError", "Capacity error",
                                "Index error", "Cancelled", "Unknown error", "NotImplemented"};
  if (ok()) return "OK";
  const int c = static_cast<int>(code_);
  return std::string(c >= 0 && c <= 10 ? names[c] : "Error") + ": " + msg_;
}

// ---- types ----------------------------------------------------------------------------------------
int DataType::bit_width() const {
  switch (id_) {
    case Type::BOOL: return 1;
    case Type::UINT8: case Type::INT8: return 8;
    case Type::UINT16: case Type::INT16: case Type::HALF_FLOAT: return 16;
    case Type::UINT32: case Type::INT32: case Type::FLOAT: return 32;
    case Type::UINT64: case Type::INT64: case Type::DOUBLE: return 64;
    default: return -1;
  }
}
std::string DataType::ToString() const {
  static const char* names[] = {"null", "bool", "uint8", "int8", "uint16", "int16", "uint32", "int32", "uint64", "int64",
                                "halffloat", "float", "double", "struct", "dictionary", "binary", "string", "list", "large_binary",
                                "large_string", "fixed_size_binary", "decimal128", "decimal256", "large_list", "fixed_size_list", "map",
                                "extension"};
  return names[id_];
}
std::shared_ptr<DataType> TypeFromId(Type::type id) {
  static std::shared_ptr<DataType> types[Type::MAX_ID];
  static std::once_flag once;
  std::call_once(once, [] { for (int i = 0; i < Type::MAX_ID; ++i) types[i] = std::make_shared<DataType>(static_cast<Type::type>(i)); });
  return types[id];
}
std::shared_ptr<DataType> large_binary() { return TypeFromId(Type::LARGE_BINARY); }
std::shared_ptr<DataType> large_utf8() { return TypeFromId(Type::LARGE_STRING); }
std::shared_ptr<DataType> fixed_size_binary(int byte_width) {
  auto t = std::make_shared<DataType>(Type::FIXED_SIZE_BINARY);
  t->set_byte_width(byte_width);
  return t;
}
std::shared_ptr<DataType> decimal128(int, int) { auto t = std::make_shared<DataType>(Type::DECIMAL128); t->set_byte_width(16); return t; }
std::shared_ptr<DataType> decimal256(int, int) { auto t = std::make_shared<DataType>(Type::DECIMAL256); t->set_byte_width(32); return t; }
static std::shared_ptr<DataType> WithValue(Type::type id, std::shared_ptr<DataType> v) {
  auto t = std::make_shared<DataType>(id);
  t->set_value_type(std::move(v));
  return t;
}
std::shared_ptr<DataType> list(std::shared_ptr<DataType> value_type) { return WithValue(Type::LIST, std::move(value_type)); }
std::shared_ptr<DataType> large_list(std::shared_ptr<DataType> value_type) { return WithValue(Type::LARGE_LIST, std::move(value_type)); }
std::shared_ptr<DataType> fixed_size_list(std::shared_ptr<DataType> value_type, int list_size) {
  auto t = WithValue(Type::FIXED_SIZE_LIST, std::move(value_type));
  t->set_list_size(list_size);
  return t;
}
std::shared_ptr<DataType> struct_(std::vector<std::pair<std::string, std::shared_ptr<DataType>>> fields) {
  auto t = std::make_shared<DataType>(Type::STRUCT);
  t->set_fields(std::move(fields));
  return t;
}
std::shared_ptr<DataType> map(std::shared_ptr<DataType> key_type, std::shared_ptr<DataType> item_type) {
  return WithValue(Type::MAP, struct_({{"key", std::move(key_type)}, {"value", std::move(item_type)}}));
}
std::shared_ptr<DataType> dictionary(std::shared_ptr<DataType> index_type, std::shared_ptr<DataType> value_type) {
  auto t = WithValue(Type::DICTIONARY, std::move(value_type));
  t->set_index_type(std::move(index_type));
  return t;
}
std::shared_ptr<DataType> extension(std::string name, std::shared_ptr<DataType> storage_type) {
  auto t = WithValue(Type::EXTENSION, std::move(storage_type));
  t->set_extension_name(std::move(name));
  return t;
}
std::shared_ptr<DataType> boolean() { return TypeFromId(Type::BOOL); }
std::shared_ptr<DataType> struct_() { return TypeFromId(Type::STRUCT); }
std::shared_ptr<DataType> dictionary_() { return TypeFromId(Type::DICTIONARY); }
std::shared_ptr<DataType> dictionary(std::shared_ptr<DataType> value_type) {
  auto t = std::make_shared<DataType>(Type::DICTIONARY);
  t->set_value_type(std::move(value_type));
  return t;
}
std::shared_ptr<DataType> null() { return TypeFromId(Type::NA); }
std::shared_ptr<DataType> list_() { return TypeFromId(Type::LIST); }
std::shared_ptr<DataType> binary() { return TypeFromId(Type::BINARY); }
std::shared_ptr<DataType> utf8() { return TypeFromId(Type::STRING); }
std::shared_ptr<DataType> uint8() { return TypeFromId(Type::UINT8); }
std::shared_ptr<DataType> int8() { return TypeFromId(Type::INT8); }
std::shared_ptr<DataType> uint16() { return TypeFromId(Type::UINT16); }
std::shared_ptr<DataType> int16() { return TypeFromId(Type::INT16); }
std::shared_ptr<DataType> uint32() { return TypeFromId(Type::UINT32); }
std::shared_ptr<DataType> int32() { return TypeFromId(Type::INT32); }
std::shared_ptr<DataType> uint64() { return TypeFromId(Type::UINT64); }
std::shared_ptr<DataType> int64() { return TypeFromId(Type::INT64); }
std::shared_ptr<DataType> float32() { return TypeFromId(Type::FLOAT); }
std::shared_ptr<DataType> float64() { return TypeFromId(Type::DOUBLE); }
bool is_signed_integer(Type::type id) { return id == Type::INT8 || id == Type::INT16 || id == Type::INT32 || id == Type::INT64; }
bool is_integer(Type::type id) { return id >= Type::UINT8 && id <= Type::INT64; }
bool is_floating(Type::type id) { return id == Type::FLOAT || id == Type::DOUBLE; }

// ---- memory ---------------------------------------------------------------------------------------
Result<std::shared_ptr<MemoryPool>> MemoryPool::Make(int device, void* stream) {
  auto pool = std::shared_ptr<MemoryPool>(new MemoryPool());
  const int rc = amdarrow_ctx_create(device, stream, /*private_stream=*/0, &pool->ctx_);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return pool;
}
MemoryPool::~MemoryPool() {
  if (ctx_) amdarrow_ctx_destroy(ctx_);
}
int64_t MemoryPool::bytes_allocated() const { return amdarrow_bytes_allocated(ctx_); }
int64_t MemoryPool::max_memory() const { return amdarrow_max_memory(ctx_); }
void MemoryPool::ReleaseUnused() { (void)amdarrow_pool_trim(ctx_, nullptr); }
int64_t MemoryPool::cached_bytes() const { return amdarrow_pool_cached_bytes(ctx_); }

Buffer::~Buffer() {
  if (owned_ && pool_ && data_) amdarrow_free(pool_->ctx(), data_);
}
std::shared_ptr<Buffer> Buffer::WrapHost(const void* data, int64_t size) {
  return std::make_shared<Buffer>(const_cast<void*>(data), size, nullptr, false);
}
Result<std::shared_ptr<Buffer>> AllocateBuffer(int64_t nbytes, const std::shared_ptr<MemoryPool>& pool) {
  if (!pool) return Status::Invalid("AllocateBuffer: no device memory pool (this backend has no CPU path)");
  void* p = nullptr;
  const int rc = amdarrow_malloc(pool->ctx(), nbytes, &p);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return std::make_shared<Buffer>(p, nbytes, pool, true);
}
Result<std::shared_ptr<Buffer>> AllocateBitmap(int64_t nbits, const std::shared_ptr<MemoryPool>& pool) {
  const int64_t nbytes = (nbits + 7) / 8;
  AMD_ASSIGN_OR_RAISE(auto buf, AllocateBuffer(nbytes, pool));
  if (nbytes > 0) {
    // the pool rounds capacity up to 64 bytes, so zeroing whole 8-byte words is in bounds
    const int rc = amdarrow_memset(pool->ctx(), buf->mutable_data(), 0, ((nbytes + 7) / 8) * 8);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  return buf;
}

// ---- ArrayData --------------------------------------------------------------------------------------
std::shared_ptr<ArrayData> ArrayData::Make(std::shared_ptr<DataType> type, int64_t length,
                                           std::vector<std::shared_ptr<Buffer>> buffers, int64_t null_count, int64_t offset) {
  auto d = std::make_shared<ArrayData>();
  d->type = std::move(type);
  d->length = length;
  d->buffers = std::move(buffers);
  if (d->buffers.size() < 2) d->buffers.resize(2);
  d->null_count = d->buffers[0] ? null_count : 0;
  d->offset = offset;
  return d;
}
std::shared_ptr<ArrayData> ArrayData::Slice(int64_t off, int64_t len) const {
  auto d = std::make_shared<ArrayData>(*this);
  d->offset = offset + off;
  d->length = len;
  // array/data.cc: a slice keeps 0, otherwise the count becomes unknown
  d->null_count = (null_count == 0 || !buffers[0]) ? 0 : (len == length ? null_count : kUnknownNullCount);
  return d;
}
amdarrow_array ArrayData::abi() const {
  amdarrow_array a;
  a.values = buffers[1] ? buffers[1]->data() : nullptr;
  a.validity = buffers[0] ? buffers[0]->data() : nullptr;
  a.length = length;
  a.offset = offset;
  a.type_id = static_cast<int32_t>(type->id());
  a.null_count = null_count;
  return a;
}
static Status CopyBitsToHost(const std::shared_ptr<MemoryPool>& pool, const uint8_t* dev, int64_t off, int64_t n, uint8_t* out);
amdarrow_var_array ArrayData::var_abi() const {
  amdarrow_var_array a;
  const Type::type id = type->id();
  a.offsets = buffers.size() > 1 && buffers[1] ? buffers[1]->data() : nullptr;
  a.offset_bits = (id == Type::LARGE_BINARY || id == Type::LARGE_STRING || id == Type::LARGE_LIST) ? 64 : 32;
  a.data = buffers.size() > 2 && buffers[2] ? buffers[2]->data() : nullptr;
  a.validity = buffers[0] ? buffers[0]->data() : nullptr;
  a.length = length;
  a.offset = offset;
  a.null_count = null_count;
  return a;
}
amdarrow_binary_array ArrayData::binary_abi() const {
  amdarrow_binary_array a;
  a.offsets = buffers.size() > 1 && buffers[1] ? reinterpret_cast<const int32_t*>(buffers[1]->data()) : nullptr;
  a.data = buffers.size() > 2 && buffers[2] ? buffers[2]->data() : nullptr;
  a.validity = buffers[0] ? buffers[0]->data() : nullptr;
  a.length = length;
  a.offset = offset;
  a.null_count = null_count;
  return a;
}
Result<std::shared_ptr<ArrayData>> BinaryFromHost(const std::shared_ptr<MemoryPool>& pool, std::shared_ptr<DataType> type,
                                                  const std::vector<std::string>& items, const std::vector<bool>& valid) {
  const int64_t n = static_cast<int64_t>(items.size());
  std::vector<int32_t> offsets(n + 1, 0);
  std::string bytes;
  std::vector<uint8_t> bits((n + 7) / 8 + 1, 0);
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; ++i) {
    const bool ok = valid.empty() || valid[i];
    if (ok) { bytes += items[i]; bits[i / 8] |= uint8_t(1u << (i % 8)); } else { ++nulls; }
    offsets[i + 1] = static_cast<int32_t>(bytes.size());
  }
  AMD_ASSIGN_OR_RAISE(auto obuf, AllocateBuffer(4 * (n + 1), pool));
  AMD_ASSIGN_OR_RAISE(auto dbuf, AllocateBuffer(static_cast<int64_t>(bytes.size()), pool));
  int rc = amdarrow_memcpy_h2d(pool->ctx(), obuf->mutable_data(), offsets.data(), 4 * (n + 1));
  if (rc == AMDARROW_OK && !bytes.empty()) rc = amdarrow_memcpy_h2d(pool->ctx(), dbuf->mutable_data(), bytes.data(), static_cast<int64_t>(bytes.size()));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  std::shared_ptr<Buffer> nbuf;
  if (nulls > 0) {
    AMD_ASSIGN_OR_RAISE(nbuf, AllocateBuffer((n + 7) / 8, pool));
    rc = amdarrow_memcpy_h2d(pool->ctx(), nbuf->mutable_data(), bits.data(), (n + 7) / 8);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  auto d = ArrayData::Make(std::move(type), n, {nbuf, obuf}, nulls);
  d->buffers.push_back(dbuf);
  return d;
}
Result<std::vector<std::pair<std::string, bool>>> BinaryToHost(const ArrayData& a, const std::shared_ptr<MemoryPool>& pool) {
  std::vector<std::pair<std::string, bool>> out;
  if (a.length == 0) return out;
  std::vector<int32_t> offsets(a.length + 1);
  int rc = amdarrow_memcpy_d2h(pool->ctx(), offsets.data(), a.buffers[1]->data() + 4 * a.offset, 4 * (a.length + 1));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  std::string bytes(static_cast<size_t>(offsets[a.length] - offsets[0]), '\0');
  if (!bytes.empty()) {
    rc = amdarrow_memcpy_d2h(pool->ctx(), &bytes[0], a.buffers[2]->data() + offsets[0], static_cast<int64_t>(bytes.size()));
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  std::vector<uint8_t> bits((a.length + 7) / 8 + 1, 0xFF);
  if (a.buffers[0]) AMD_RETURN_NOT_OK(CopyBitsToHost(pool, a.buffers[0]->data(), a.offset, a.length, bits.data()));
  for (int64_t i = 0; i < a.length; ++i) {
    const bool ok = (bits[i / 8] >> (i % 8)) & 1;
    out.emplace_back(ok ? bytes.substr(offsets[i] - offsets[0], offsets[i + 1] - offsets[i]) : std::string(), ok);
  }
  return out;
}
Result<std::shared_ptr<ArrayData>> ArrayFromHost(const std::shared_ptr<MemoryPool>& pool, std::shared_ptr<DataType> type,
                                                 const void* values, int64_t length, const uint8_t* validity_bits,
                                                 int64_t null_count) {
  const int bw = type->bit_width();
  const int64_t vbytes = bw == 1 ? (length + 7) / 8 : length * bw / 8;
  AMD_ASSIGN_OR_RAISE(auto vbuf, AllocateBuffer(vbytes, pool));
  if (vbytes) {
    const int rc = amdarrow_memcpy_h2d(pool->ctx(), vbuf->mutable_data(), values, vbytes);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  std::shared_ptr<Buffer> nbuf;
  if (validity_bits) {
    AMD_ASSIGN_OR_RAISE(nbuf, AllocateBuffer((length + 7) / 8, pool));
    if (length) {
      const int rc = amdarrow_memcpy_h2d(pool->ctx(), nbuf->mutable_data(), validity_bits, (length + 7) / 8);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
  }
  return ArrayData::Make(std::move(type), length, {nbuf, vbuf}, null_count);
}
// Bits [off, off+n) of a device bitmap -> host bits at offset 0.
static Status CopyBitsToHost(const std::shared_ptr<MemoryPool>& pool, const uint8_t* dev, int64_t off, int64_t n, uint8_t* out) {
  const int64_t first = off / 8, nbytes = (off % 8 + n + 7) / 8;
  std::vector<uint8_t> raw(nbytes + 1, 0);
  const int rc = amdarrow_memcpy_d2h(pool->ctx(), raw.data(), dev + first, nbytes);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  const int shift = static_cast<int>(off % 8);
  for (int64_t b = 0; b < (n + 7) / 8; ++b)
    out[b] = static_cast<uint8_t>(shift == 0 ? raw[b] : ((raw[b] >> shift) | (raw[b + 1] << (8 - shift))));
  return Status::OK();
}
Status CopyValuesToHost(const ArrayData& a, const std::shared_ptr<MemoryPool>& pool, void* out_values,
                        uint8_t* out_validity_bits) {
  const int bw = a.type->bit_width();
  if (a.length && out_values) {
    if (bw == 1) {
      AMD_RETURN_NOT_OK(CopyBitsToHost(pool, a.buffers[1]->data(), a.offset, a.length, static_cast<uint8_t*>(out_values)));
    } else {
      const int rc = amdarrow_memcpy_d2h(pool->ctx(), out_values, a.buffers[1]->data() + a.offset * bw / 8, a.length * bw / 8);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
  }
  if (out_validity_bits && a.length) {
    if (a.buffers[0]) AMD_RETURN_NOT_OK(CopyBitsToHost(pool, a.buffers[0]->data(), a.offset, a.length, out_validity_bits));
    else memset(out_validity_bits, 0xFF, (a.length + 7) / 8);
  }
  return Status::OK();
}

ChunkedArray::ChunkedArray(std::vector<std::shared_ptr<ArrayData>> chunks, std::shared_ptr<DataType> type)
    : chunks_(std::move(chunks)), type_(std::move(type)) {
  for (const auto& c : chunks_) {
    length_ += c->length;
    if (null_count_ >= 0) null_count_ = c->null_count < 0 ? kUnknownNullCount : null_count_ + c->null_count;
  }
}

std::string ValueDescr::ToString() const {
  return std::string(shape == ARRAY ? "array[" : shape == SCALAR ? "scalar[" : "any[") + (type ? type->ToString() : "?") + "]";
}
std::shared_ptr<DataType> Datum::type() const {
  switch (kind_) {
    case ARRAY: return array_->type;
    case CHUNKED_ARRAY: return chunked_->type();
    case SCALAR: return scalar_->type;
    default: return nullptr;
  }
}
int64_t Datum::length() const {
  switch (kind_) {
    case ARRAY: return array_->length;
    case CHUNKED_ARRAY: return chunked_->length();
    case SCALAR: return 1;
    case RECORD_BATCH: return batch_->num_rows();  // datum.cc:108-124
    case TABLE: return table_->num_rows();
    default: return -1;
  }
}

Result<std::shared_ptr<ArrayData>> Concatenate(const std::vector<std::shared_ptr<ArrayData>>& arrays,
                                               const std::shared_ptr<MemoryPool>& pool) {
  if (arrays.empty()) return Status::Invalid("Must pass at least one array");  // concatenate.cc:428-430
  std::vector<amdarrow_array> abi;
  int64_t total = 0, nulls = 0;
  for (const auto& a : arrays) {
    if (!a->type->Equals(*arrays[0]->type))  // :436-441
      return Status::Invalid("arrays to be concatenated must be identically typed, but " + arrays[0]->type->ToString() +
                             " and " + a->type->ToString() + " were encountered.");
    abi.push_back(a->abi());
    total += a->length;
    const int64_t nc = a->buffers[0] ? a->null_count : 0;
    nulls = (nulls < 0 || nc < 0) ? -1 : nulls + nc;
  }
  if (is_binary_like(arrays[0]->type->id())) {  // ConcatenateImpl::Visit(BinaryType) :203-214
    std::vector<amdarrow_binary_array> babi;
    for (const auto& a : arrays) babi.push_back(a->binary_abi());
    int64_t rows = 0, nbytes = 0;
    int rc = amdarrow_binary_concatenate_size(pool->ctx(), babi.data(), static_cast<int>(babi.size()), &rows, &nbytes);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    AMD_ASSIGN_OR_RAISE(auto obuf, AllocateBuffer(4 * (rows + 1), pool));
    AMD_ASSIGN_OR_RAISE(auto dbuf, AllocateBuffer(nbytes, pool));
    std::shared_ptr<Buffer> vb;
    if (nulls != 0) { AMD_ASSIGN_OR_RAISE(vb, AllocateBitmap(rows, pool)); }
    amdarrow_out bo;
    rc = amdarrow_binary_concatenate(pool->ctx(), babi.data(), static_cast<int>(babi.size()), reinterpret_cast<int32_t*>(obuf->mutable_data()),
                                     dbuf->mutable_data(), vb ? vb->mutable_data() : nullptr, &bo);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    auto d = ArrayData::Make(arrays[0]->type, bo.length, {vb, obuf}, bo.null_count);
    d->buffers.push_back(dbuf);
    return d;
  }
  const int bw = arrays[0]->type->bit_width();
  std::shared_ptr<Buffer> vbuf, nbuf;
  if (bw == 1) { AMD_ASSIGN_OR_RAISE(vbuf, AllocateBitmap(total, pool)); }
  else { AMD_ASSIGN_OR_RAISE(vbuf, AllocateBuffer(total * bw / 8, pool)); }
  if (nulls != 0) { AMD_ASSIGN_OR_RAISE(nbuf, AllocateBitmap(total, pool)); }  // :190-192
  amdarrow_out o;
  const int rc = amdarrow_concatenate(pool->ctx(), abi.data(), static_cast<int>(abi.size()), vbuf->mutable_data(),
                                      nbuf ? nbuf->mutable_data() : nullptr, &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return ArrayData::Make(arrays[0]->type, o.length, {nbuf, vbuf}, o.null_count);
}

ValueDescr Datum::descr() const {
  return kind_ == SCALAR ? ValueDescr::Scalar(type()) : ValueDescr::Array(type());
}

namespace compute {

namespace internal {
std::vector<std::vector<std::shared_ptr<ArrayData>>> RechunkArraysConsistently(
    const std::vector<std::vector<std::shared_ptr<ArrayData>>>& groups) {
  if (groups.size() <= 1) return groups;
  int64_t total = 0;
  for (const auto& a : groups.front()) total += a->length;
  if (total == 0) return groups;
  std::vector<std::vector<std::shared_ptr<ArrayData>>> out(groups.size());
  std::vector<size_t> cur(groups.size(), 0);
  std::vector<int64_t> off(groups.size(), 0);
  for (int64_t start = 0; start < total;) {
    int64_t len = std::numeric_limits<int64_t>::max();
    for (size_t g = 0; g < groups.size(); ++g) {
      while (off[g] == groups[g][cur[g]]->length) { ++cur[g]; off[g] = 0; }  // done and 0-length arrays
      len = std::min(len, groups[g][cur[g]]->length - off[g]);
    }
    for (size_t g = 0; g < groups.size(); ++g) {
      const auto& a = groups[g][cur[g]];
      out[g].push_back(off[g] == 0 && len == a->length ? a : a->Slice(off[g], len));
      off[g] += len;
    }
    start += len;
  }
  return out;
}
}  // namespace internal

ExecContext::ExecContext(std::shared_ptr<MemoryPool> pool, FunctionRegistry* func_registry)
    : pool_(std::move(pool)), func_registry_(func_registry ? func_registry : GetFunctionRegistry()) {}

// ---- ExecBatch / ExecBatchIterator -----------------------------------------------------------------
Result<ExecBatch> ExecBatch::Make(std::vector<Datum> values) {  // exec.cc:60-84
  if (values.empty()) return Status::Invalid("Cannot infer ExecBatch length without at least one value");
  int64_t length = -1;
  for (const auto& v : values) {
    if (v.is_scalar()) {
      if (length == -1) length = 1;
      continue;
    }
    if (!v.is_array()) return Status::Invalid("ExecBatch::Make: only ARRAY and SCALAR values");
    if (length == -1 || length == 1) {
      length = v.length();
      continue;
    }
    if (length != v.length()) return Status::Invalid("Arrays used to construct an ExecBatch must have equal length");
  }
  return ExecBatch(std::move(values), length);
}
std::vector<ValueDescr> ExecBatch::GetDescriptors() const {
  std::vector<ValueDescr> d;
  for (const auto& v : values) d.push_back(v.descr());
  return d;
}

namespace detail {
Result<std::unique_ptr<ExecBatchIterator>> ExecBatchIterator::Make(std::vector<Datum> args, int64_t max_chunksize) {
  // exec.cc:158-196: every array-like argument must have the same length
  int64_t length = 1;
  bool length_set = false;
  for (const auto& arg : args) {
    if (!(arg.is_arraylike() || arg.is_scalar()))
      return Status::Invalid("ExecBatchIterator only works with Scalar, Array, and ChunkedArray arguments");
    if (arg.is_scalar()) continue;
    if (!length_set) {
      length = arg.length();
      length_set = true;
    } else if (arg.length() != length) {
      return Status::Invalid("Array arguments must all be the same length");  // exec.cc:190-192
    }
  }
  auto it = std::unique_ptr<ExecBatchIterator>(new ExecBatchIterator());
  it->args_ = std::move(args);
  it->chunk_indexes_.assign(it->args_.size(), 0);
  it->chunk_positions_.assign(it->args_.size(), 0);
  it->length_ = length;
  it->max_chunksize_ = std::min(length, max_chunksize);
  return it;
}
bool ExecBatchIterator::Next(ExecBatch* batch) {  // exec.cc:198-252
  if (position_ == length_) return false;
  int64_t iteration_size = std::min(length_ - position_, max_chunksize_);
  // first pass: the largest slice available in every chunked argument
  for (size_t i = 0; i < args_.size() && iteration_size > 0; ++i) {
    if (args_[i].kind() != Datum::CHUNKED_ARRAY) continue;
    const ChunkedArray& arg = *args_[i].chunked_array();
    while (chunk_indexes_[i] < arg.num_chunks()) {
      const ArrayData& chunk = *arg.chunk(chunk_indexes_[i]);
      if (chunk_positions_[i] == chunk.length) {  // chunk exhausted (or empty): advance
        ++chunk_indexes_[i];
        chunk_positions_[i] = 0;
        continue;
      }
      iteration_size = std::min(chunk.length - chunk_positions_[i], iteration_size);
      break;
    }
  }
  batch->values.resize(args_.size());
  batch->length = iteration_size;
  for (size_t i = 0; i < args_.size(); ++i) {
    if (args_[i].is_scalar()) {
      batch->values[i] = args_[i];
    } else if (args_[i].is_array()) {
      batch->values[i] = Datum(args_[i].array()->Slice(position_, iteration_size));
    } else {
      const ChunkedArray& carr = *args_[i].chunked_array();
      const auto& chunk = carr.chunk(chunk_indexes_[i]);
      batch->values[i] = Datum(chunk->Slice(chunk_positions_[i], iteration_size));
      chunk_positions_[i] += iteration_size;
    }
  }
  position_ += iteration_size;
  return true;
}
}  // namespace detail

// ---- signatures -------------------------------------------------------------------------------------
InputType InputType::Exact(Type::type id, ValueDescr::Shape shape) {
  InputType t;
  t.matches = [id, shape](const ValueDescr& d) { return d.type && d.type->id() == id && (shape == ValueDescr::ANY || shape == d.shape); };
  t.repr = TypeFromId(id)->ToString();
  return t;
}
InputType InputType::Primitive() {
  InputType t;
  t.matches = [](const ValueDescr& d) { return d.type && d.type->bit_width() > 0 && d.type->id() != Type::HALF_FLOAT; };
  t.repr = "primitive";
  return t;
}
InputType InputType::Integer() {
  InputType t;
  t.matches = [](const ValueDescr& d) { return d.type && is_integer(d.type->id()); };
  t.repr = "integer";
  return t;
}
InputType InputType::Any() {
  InputType t;
  t.matches = [](const ValueDescr&) { return true; };
  t.repr = "any";
  return t;
}
bool KernelSignature::MatchesInputs(const std::vector<ValueDescr>& args) const {
  if (args.size() != in_types.size()) return false;
  for (size_t i = 0; i < args.size(); ++i)
    if (!in_types[i].matches(args[i])) return false;
  return true;
}

// ---- Function ------------------------------------------------------------------------------------------
Status Function::CheckArity(int passed) const {  // function.cc:41-56
  if (arity_.is_varargs && passed < arity_.num_args)
    return Status::Invalid("VarArgs function " + name_ + " needs at least " + std::to_string(arity_.num_args) +
                           " arguments but kernel accepts only " + std::to_string(passed));
  if (!arity_.is_varargs && passed != arity_.num_args)
    return Status::Invalid("Function " + name_ + " accepts " + std::to_string(arity_.num_args) + " arguments but attempted to look up kernel(s) with " +
                           std::to_string(passed));
  return Status::OK();
}
Result<Datum> Function::Execute(const std::vector<Datum>&, const FunctionOptions*, ExecContext*) const {
  return Status::NotImplemented("Function " + name_ + " cannot be executed");
}

static std::string DescrsToString(const std::vector<ValueDescr>& v) {
  std::string s = "(";
  for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + v[i].ToString();
  return s + ")";
}
template <typename K>
static Result<const K*> Dispatch(const std::string& name, const std::vector<K>& kernels, const std::vector<ValueDescr>& values) {
  for (const auto& k : kernels)
    if (k.signature.MatchesInputs(values)) return &k;
  // function.cc:112-125 DispatchExactImpl -> NoMatchingKernel
  return Status::NotImplemented("Function " + name + " has no kernel matching input types " + DescrsToString(values));
}
Status ScalarFunction::AddKernel(ScalarKernel kernel) {
  if (static_cast<int>(kernel.signature.in_types.size()) != arity_.num_args && !arity_.is_varargs)
    return Status::Invalid("In function " + name_ + ": number of argument types != function arity");
  kernels_.push_back(std::move(kernel));
  return Status::OK();
}
Status VectorFunction::AddKernel(VectorKernel kernel) {
  if (static_cast<int>(kernel.signature.in_types.size()) != arity_.num_args && !arity_.is_varargs)
    return Status::Invalid("In function " + name_ + ": number of argument types != function arity");
  kernels_.push_back(std::move(kernel));
  return Status::OK();
}
Status HashAggregateFunction::AddKernel(HashAggregateKernel kernel) {
  kernels_.push_back(std::move(kernel));
  return Status::OK();
}
Result<const ScalarKernel*> ScalarFunction::DispatchExact(const std::vector<ValueDescr>& v) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(v.size())));
  return Dispatch(name_, kernels_, v);
}
Result<const VectorKernel*> VectorFunction::DispatchExact(const std::vector<ValueDescr>& v) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(v.size())));
  return Dispatch(name_, kernels_, v);
}
Status ScalarAggregateFunction::AddKernel(ScalarAggregateKernel kernel) {
  kernels_.push_back(std::move(kernel));
  return Status::OK();
}
Result<const ScalarAggregateKernel*> ScalarAggregateFunction::DispatchExact(const std::vector<ValueDescr>& v) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(v.size())));
  return Dispatch(name_, kernels_, v);
}
Result<const HashAggregateKernel*> HashAggregateFunction::DispatchExact(const std::vector<ValueDescr>& v) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(v.size())));
  return Dispatch(name_, kernels_, v);
}

// ---- implicit promotion and casts (codegen_internal.cc:190-268, cast.cc:120-243, scalar_cast_numeric.cc) ----
namespace internal {
static bool IsSignedInt(Type::type id) { return id == Type::INT8 || id == Type::INT16 || id == Type::INT32 || id == Type::INT64; }
std::shared_ptr<DataType> CommonNumeric(const std::vector<ValueDescr>& descrs) {  // :220-268
  for (const auto& d : descrs) {
    const Type::type id = d.type->id();
    if (!is_floating(id) && !is_integer(id)) return nullptr;  // half floats have no arithmetic either
  }
  for (const auto& d : descrs) if (d.type->id() == Type::DOUBLE) return float64();
  for (const auto& d : descrs) if (d.type->id() == Type::FLOAT) return float32();
  int width_signed = 0, width_unsigned = 0;
  for (const auto& d : descrs) {
    int* w = IsSignedInt(d.type->id()) ? &width_signed : &width_unsigned;
    *w = std::max(*w, d.type->bit_width());
  }
  if (width_signed == 0) {
    if (width_unsigned >= 64) return uint64();
    if (width_unsigned == 32) return uint32();
    if (width_unsigned == 16) return uint16();
    return int8();  // as the reference has it (:252); unreachable through DispatchBest, equal types match exactly
  }
  if (width_signed <= width_unsigned) {  // the next power of two above the unsigned width
    width_signed = 8;
    while (width_signed < width_unsigned + 1) width_signed *= 2;
  }
  if (width_signed >= 64) return int64();
  if (width_signed == 32) return int32();
  if (width_signed == 16) return int16();
  return int8();
}
void ReplaceNullWithOtherType(std::vector<ValueDescr>* descrs) {  // :197-210
  if (descrs->size() != 2) return;
  if ((*descrs)[0].type->id() == Type::NA) { (*descrs)[0].type = (*descrs)[1].type; return; }
  if ((*descrs)[1].type->id() == Type::NA) (*descrs)[1].type = (*descrs)[0].type;
}
void EnsureDictionaryDecoded(std::vector<ValueDescr>* descrs) {  // :190-196
  for (auto& d : *descrs)
    if (d.type->id() == Type::DICTIONARY && d.type->value_type()) d.type = d.type->value_type();
}
void ReplaceTypes(const std::shared_ptr<DataType>& type, std::vector<ValueDescr>* descrs) {
  for (auto& d : *descrs) d.type = type;
}
Result<std::vector<Datum>> Cast(std::vector<Datum> datums, const std::vector<ValueDescr>& descrs, ExecContext* ctx) {
  for (size_t i = 0; i < datums.size(); ++i) {
    const ValueDescr have = datums[i].descr();
    if (have.type->id() == descrs[i].type->id() && have.type->id() != Type::DICTIONARY) continue;
    if (have.type->id() == Type::DICTIONARY && descrs[i].type->id() == Type::DICTIONARY) continue;
    if ((have.shape == ValueDescr::SCALAR) != (descrs[i].shape == ValueDescr::SCALAR))
      return Status::NotImplemented("casting between Datum shapes");
    AMD_ASSIGN_OR_RAISE(datums[i], compute::Cast(datums[i], CastOptions::Safe(descrs[i].type), ctx));
  }
  return datums;
}
}  // namespace internal

namespace {
struct IntRange { __int128 lo, hi; };
IntRange RangeOf(Type::type id) {
  const int w = DataType(id).bit_width();
  if (internal::IsSignedInt(id)) return {-(static_cast<__int128>(1) << (w - 1)), (static_cast<__int128>(1) << (w - 1)) - 1};
  return {0, (static_cast<__int128>(1) << w) - 1};
}
std::string Int128ToString(__int128 v) {
  return v < 0 ? std::to_string(static_cast<long long>(v)) : std::to_string(static_cast<unsigned long long>(v));
}
// One value through the same conversion the device kernel applies (static_cast semantics, one rounding).
Result<std::shared_ptr<Scalar>> CastScalar(const Scalar& s, const std::shared_ptr<DataType>& to, const CastOptions& o) {
  const Type::type from = s.type->id(), tid = to->id();
  if (from == Type::NA || !s.is_valid) return Scalar::MakeNull(to);
  const bool from_int = is_integer(from), to_int = is_integer(tid);
  if ((!from_int && !is_floating(from)) || (!to_int && !is_floating(tid)))
    return Status::NotImplemented("Unsupported cast from " + s.type->ToString() + " to " + to->ToString());
  if (!from_int && to_int) return Status::NotImplemented("cast: floating point to integer");
  auto out = std::make_shared<Scalar>();
  out->type = to;
  if (from_int) {
    const int w = s.type->bit_width();
    __int128 v = static_cast<__int128>(s.bits & (w == 64 ? ~0ull : ((1ull << w) - 1)));
    if (internal::IsSignedInt(from) && (v >> (w - 1)) & 1) v -= static_cast<__int128>(1) << w;
    IntRange src = RangeOf(from), dst = src;
    if (to_int) dst = RangeOf(tid);
    else if (w >= (tid == Type::FLOAT ? 32 : 64)) { dst.hi = static_cast<__int128>(1) << (tid == Type::FLOAT ? 24 : 53); dst.lo = -dst.hi; }
    const __int128 lo = std::max(src.lo, dst.lo), hi = std::min(src.hi, dst.hi);
    const bool checked = to_int ? !o.allow_int_overflow : !o.allow_float_truncate;
    if (checked && (v < lo || v > hi))
      return Status::Invalid("Integer value " + Int128ToString(v) + " not in range: " + Int128ToString(lo) + " to " + Int128ToString(hi));
    if (to_int) {
      out->bits = static_cast<uint64_t>(v);
      if (to->bit_width() < 64) out->bits &= (1ull << to->bit_width()) - 1;
    } else if (tid == Type::FLOAT) {
      const float f = from == Type::UINT64 ? static_cast<float>(static_cast<uint64_t>(v)) : static_cast<float>(static_cast<int64_t>(v));
      memcpy(&out->bits, &f, 4);
    } else {
      const double d = from == Type::UINT64 ? static_cast<double>(static_cast<uint64_t>(v)) : static_cast<double>(static_cast<int64_t>(v));
      memcpy(&out->bits, &d, 8);
    }
    return out;
  }
  double d;
  if (from == Type::FLOAT) { float f; memcpy(&f, &s.bits, 4); d = f; } else memcpy(&d, &s.bits, 8);
  if (tid == Type::FLOAT) { const float f = static_cast<float>(d); memcpy(&out->bits, &f, 4); } else memcpy(&out->bits, &d, 8);
  return out;
}
Result<std::shared_ptr<ArrayData>> CastArray(const ArrayData& a, const std::shared_ptr<DataType>& to, const CastOptions& o, ExecContext* ctx) {
  const auto& pool = ctx->memory_pool();
  const Type::type from = a.type->id(), tid = to->id();
  if (from == Type::DICTIONARY) {  // scalar_cast_dictionary.cc: unpack = take(dictionary, indices), then on to `to`
    if (!a.dictionary) return Status::Invalid("dictionary array without a dictionary");
    auto indices = std::make_shared<ArrayData>(a);
    indices->type = int32();
    indices->dictionary = nullptr;
    AMD_ASSIGN_OR_RAISE(Datum decoded, Take(Datum(a.dictionary), Datum(indices), TakeOptions::NoBoundsCheck(), ctx));
    if (decoded.type()->id() == tid) return decoded.array();
    return CastArray(*decoded.array(), to, o, ctx);
  }
  if (to->bit_width() < 8 || tid == Type::HALF_FLOAT)
    return Status::NotImplemented("Unsupported cast from " + a.type->ToString() + " to " + to->ToString());
  AMD_ASSIGN_OR_RAISE(auto values, AllocateBuffer(std::max<int64_t>(a.length, 1) * (to->bit_width() / 8), pool));
  if (from == Type::NA) {  // CastFromNull (scalar_cast_internal.cc): an all-null array of the target type
    AMD_ASSIGN_OR_RAISE(auto bitmap, AllocateBitmap(std::max<int64_t>(a.length, 1), pool));
    int rc = amdarrow_memset(pool->ctx(), values->mutable_data(), 0, values->size());
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    return ArrayData::Make(to, a.length, {bitmap, values}, a.length);
  }
  if (a.type->bit_width() < 8 || from == Type::HALF_FLOAT || from > Type::DOUBLE)
    return Status::NotImplemented("Unsupported cast from " + a.type->ToString() + " to " + to->ToString());
  const amdarrow_array abi = a.abi();
  const bool safe = is_integer(tid) ? !o.allow_int_overflow : !o.allow_float_truncate;
  const int rc = amdarrow_cast_numeric(pool->ctx(), &abi, static_cast<int>(tid), safe ? 1 : 0, values->mutable_data());
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  // the validity buffer is shared (ZeroCopyCastExec / the executors' pre-allocated validity of a unary INTERSECTION
  // kernel, exec.cc:399-420): it keeps its bit offset, so the new values are re-based through ArrayData::offset
  if (!a.buffers[0] || a.offset == 0) return ArrayData::Make(to, a.length, {a.buffers[0], values}, a.buffers[0] ? a.null_count : 0);
  AMD_ASSIGN_OR_RAISE(auto bitmap, AllocateBitmap(a.length, pool));
  amdarrow_out vo;
  const int rc2 = amdarrow_validity_test(pool->ctx(), 0 /*is_valid*/, &abi, bitmap->mutable_data(), &vo);
  if (rc2 != AMDARROW_OK) return Status::FromAbi(rc2);
  return ArrayData::Make(to, a.length, {bitmap, values}, a.null_count);
}
}  // namespace

Result<Datum> Cast(const Datum& value, const CastOptions& options, ExecContext* ctx) {
  if (!options.to_type) return Status::Invalid("Cast requires that options.to_type be set");  // cast.cc:140-142
  const auto& to = options.to_type;
  switch (value.kind()) {
    case Datum::SCALAR: {
      AMD_ASSIGN_OR_RAISE(auto s, CastScalar(*value.scalar(), to, options));
      return Datum(s);
    }
    case Datum::ARRAY: {
      if (value.type()->id() == to->id() && to->id() != Type::DICTIONARY) return value;  // cast.cc:143-145
      AMD_ASSIGN_OR_RAISE(auto a, CastArray(*value.array(), to, options, ctx));
      return Datum(a);
    }
    case Datum::CHUNKED_ARRAY: {
      if (value.type()->id() == to->id() && to->id() != Type::DICTIONARY) return value;
      std::vector<std::shared_ptr<ArrayData>> chunks;
      for (const auto& c : value.chunked_array()->chunks()) {
        AMD_ASSIGN_OR_RAISE(auto a, CastArray(*c, to, options, ctx));
        chunks.push_back(std::move(a));
      }
      return Datum(std::make_shared<ChunkedArray>(std::move(chunks), to));
    }
    default: return Status::NotImplemented("Unsupported Datum kind for cast");
  }
}
Result<Datum> Cast(const Datum& value, std::shared_ptr<DataType> to_type, const CastOptions& options, ExecContext* ctx) {
  CastOptions o = options;
  o.to_type = std::move(to_type);
  return Cast(value, o, ctx);
}

namespace internal {
namespace {
class CastMetaFunction : public MetaFunction {  // cast.cc:120-160
 public:
  CastMetaFunction() : MetaFunction("cast", Arity::Unary()) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override {
    const auto* o = static_cast<const CastOptions*>(options);
    if (o == nullptr || o->to_type == nullptr) return Status::Invalid("Cast requires that options be passed with the to_type populated");
    return compute::Cast(args[0], *o, ctx);
  }
};
}  // namespace
void RegisterScalarCast(FunctionRegistry* registry) { (void)registry->AddFunction(std::make_shared<CastMetaFunction>()); }
}  // namespace internal

static bool HaveChunkedArray(const std::vector<Datum>& args) {
  for (const auto& a : args) if (a.kind() == Datum::CHUNKED_ARRAY) return true;
  return false;
}
// exec.cc:463-480 ToChunkedArray: empty chunks are dropped
static Datum WrapResults(const std::vector<Datum>& inputs, std::vector<Datum> outputs, const std::shared_ptr<DataType>& type,
                         const std::shared_ptr<MemoryPool>& pool) {
  if (HaveChunkedArray(inputs) || outputs.size() > 1) {
    std::vector<std::shared_ptr<ArrayData>> chunks;
    for (auto& o : outputs) if (o.array()->length > 0) chunks.push_back(o.array());
    return Datum(std::make_shared<ChunkedArray>(std::move(chunks), type));
  }
  if (outputs.size() == 1) return outputs[0];
  return Datum(ArrayData::Make(type, 0, {nullptr, nullptr}, 0));
}

// ScalarExecutor (exec.cc:553-716).  Output preallocation and null propagation happen behind the
// C ABI for the kernels of this backend (they receive out buffers at offset 0), so a batch split
// by exec_chunksize yields one output chunk per batch -- what the reference does for kernels with
// can_write_into_slices == false.
Result<Datum> ScalarFunction::Execute(const std::vector<Datum>& args_in, const FunctionOptions* options, ExecContext* ctx) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(args_in.size())));
  std::vector<ValueDescr> descrs;
  bool all_scalar = true;
  for (const auto& a : args_in) {
    if (!(a.is_arraylike() || a.is_scalar())) return Status::NotImplemented("Unsupported Datum kind for scalar function " + name_);
    descrs.push_back(a.descr());
    all_scalar &= a.is_scalar();
  }
  // a dictionary array's descriptor names its value type (DictionaryType, type.h:1360) for EnsureDictionaryDecoded
  for (size_t i = 0; i < args_in.size(); ++i)
    if (descrs[i].type->id() == Type::DICTIONARY && !descrs[i].type->value_type() && args_in[i].is_array() && args_in[i].array()->dictionary)
      descrs[i].type = dictionary(args_in[i].array()->dictionary->type);
  // function.cc:175-176: the best kernel, then the arguments cast to the descriptors it settled on
  const std::vector<ValueDescr> given = descrs;
  AMD_ASSIGN_OR_RAISE(const ScalarKernel* kernel, DispatchBest(&descrs));
  if (all_scalar) {
    // exec.cc:553-716: a call whose arguments are all scalars yields a scalar.  The kernels here are array kernels, so
    // the first argument becomes a one-element array and the element of the result is boxed again.
    std::vector<Datum> lifted = args_in;
    const Scalar& s0 = *args_in[0].scalar();
    if (s0.type->bit_width() < 1) return Status::NotImplemented("Function " + name_ + ": scalar of type " + s0.type->ToString());
    const uint8_t valid_bit = s0.is_valid ? 1 : 0;
    uint64_t bits = s0.bits;
    AMD_ASSIGN_OR_RAISE(auto one, ArrayFromHost(ctx->memory_pool(), s0.type, &bits, 1, &valid_bit, s0.is_valid ? 0 : 1));
    lifted[0] = Datum(one);
    AMD_ASSIGN_OR_RAISE(Datum r, Execute(lifted, options, ctx));
    if (!r.is_array() || r.array()->length != 1) return Status::Invalid("scalar execution did not yield one value");
    const ArrayData& a = *r.array();
    if (a.type->bit_width() < 1) return Status::NotImplemented("Function " + name_ + ": scalar result of type " + a.type->ToString());
    uint64_t value = 0;
    uint8_t vbit = 1;
    AMD_RETURN_NOT_OK(CopyValuesToHost(a, ctx->memory_pool(), &value, &vbit));
    auto out = std::make_shared<Scalar>();
    out->type = a.type;
    out->is_valid = (vbit & 1) != 0;
    out->bits = a.type->bit_width() == 1 ? (value & 1) : value;
    if (!out->is_valid) out->bits = 0;
    return Datum(out);
  }
  std::vector<Datum> cast_args;
  bool changed = false;
  for (size_t i = 0; i < descrs.size(); ++i) changed |= descrs[i].type.get() != given[i].type.get();
  if (changed) { AMD_ASSIGN_OR_RAISE(cast_args, internal::Cast(args_in, descrs, ctx)); }
  const std::vector<Datum>& args = changed ? cast_args : args_in;
  AMD_ASSIGN_OR_RAISE(auto out_type, kernel->signature.out_type(descrs));
  KernelContext kctx(ctx);
  kctx.SetOptions(options ? options : default_options_);
  AMD_ASSIGN_OR_RAISE(auto it, detail::ExecBatchIterator::Make(args, ctx->exec_chunksize()));
  std::vector<Datum> outputs;
  ExecBatch batch;
  while (it->Next(&batch)) {
    Datum out;
    AMD_RETURN_NOT_OK(kernel->exec(&kctx, batch, &out));
    outputs.push_back(std::move(out));
  }
  return WrapResults(args, std::move(outputs), out_type, ctx->memory_pool());
}

// VectorExecutor (exec.cc:748-800)
Result<Datum> VectorFunction::Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(args.size())));
  std::vector<ValueDescr> descrs;
  for (const auto& a : args) {
    if (!(a.is_arraylike() || a.is_scalar())) return Status::NotImplemented("Unsupported Datum kind for vector function " + name_);
    descrs.push_back(a.descr());
  }
  AMD_ASSIGN_OR_RAISE(const VectorKernel* kernel, DispatchExact(descrs));
  AMD_ASSIGN_OR_RAISE(auto out_type, kernel->signature.out_type(descrs));
  KernelContext kctx(ctx);
  kctx.SetOptions(options ? options : default_options_);
  std::vector<Datum> outputs;
  if (kernel->can_execute_chunkwise) {
    AMD_ASSIGN_OR_RAISE(auto it, detail::ExecBatchIterator::Make(args, ctx->exec_chunksize()));
    ExecBatch batch;
    while (it->Next(&batch)) {
      Datum out;
      AMD_RETURN_NOT_OK(kernel->exec(&kctx, batch, &out));
      outputs.push_back(std::move(out));
    }
  } else {
    // PackBatchNoChunks (exec.cc:718-738): the kernel sees the arguments whole
    int64_t length = 0;
    for (const auto& a : args) length = std::max(length, a.length());
    ExecBatch batch(args, length);
    Datum out;
    AMD_RETURN_NOT_OK(kernel->exec(&kctx, batch, &out));
    if (out.kind() == Datum::CHUNKED_ARRAY) return out;
    outputs.push_back(std::move(out));
    return outputs[0];
  }
  return WrapResults(args, std::move(outputs), out_type, ctx->memory_pool());
}

// ScalarAggExecutor (exec.cc:802-903)
Result<Datum> ScalarAggregateFunction::Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(args.size())));
  std::vector<ValueDescr> descrs;
  for (const auto& a : args) {
    if (!a.is_arraylike()) return Status::NotImplemented("Unsupported Datum kind for aggregate function " + name_);
    descrs.push_back(a.descr());
  }
  AMD_ASSIGN_OR_RAISE(const ScalarAggregateKernel* kernel, DispatchExact(descrs));
  KernelContext kctx(ctx);
  const FunctionOptions* opts = options ? options : default_options_;
  kctx.SetOptions(opts);
  AMD_ASSIGN_OR_RAISE(auto state, kernel->init(&kctx, descrs[0].type, opts));
  AMD_ASSIGN_OR_RAISE(auto it, detail::ExecBatchIterator::Make(args, ctx->exec_chunksize()));
  ExecBatch batch;
  while (it->Next(&batch)) {
    AMD_ASSIGN_OR_RAISE(auto batch_state, kernel->init(&kctx, descrs[0].type, opts));  // :861-866
    AMD_RETURN_NOT_OK(kernel->consume(&kctx, batch_state.get(), batch));
    AMD_RETURN_NOT_OK(kernel->merge(&kctx, std::move(*batch_state), state.get()));
  }
  Datum out;
  AMD_RETURN_NOT_OK(kernel->finalize(&kctx, state.get(), &out));
  return out;
}

Result<Datum> HashAggregateFunction::Execute(const std::vector<Datum>&, const FunctionOptions*, ExecContext*) const {
  return Status::NotImplemented("Direct execution of HASH_AGGREGATE functions");  // function.cc:193-195
}
Result<Datum> MetaFunction::Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const {
  AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(args.size())));  // function.cc:281-291
  return ExecuteImpl(args, options ? options : default_options_, ctx);
}

// ---- registry (registry.cc:35-110) -----------------------------------------------------------------------
std::unique_ptr<FunctionRegistry> FunctionRegistry::Make() { return std::unique_ptr<FunctionRegistry>(new FunctionRegistry()); }
Status FunctionRegistry::AddFunction(std::shared_ptr<Function> function, bool allow_overwrite) {
  std::lock_guard<std::mutex> lock(mutex_);
  const std::string& name = function->name();
  if (!allow_overwrite && name_to_function_.count(name))
    return Status::KeyError("Already have a function registered with name: " + name);
  name_to_function_[name] = std::move(function);
  return Status::OK();
}
Status FunctionRegistry::AddAlias(const std::string& target_name, const std::string& source_name) {
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = name_to_function_.find(source_name);
  if (it == name_to_function_.end()) return Status::KeyError("No function registered with name: " + source_name);
  name_to_function_[target_name] = it->second;
  return Status::OK();
}
Result<std::shared_ptr<Function>> FunctionRegistry::GetFunction(const std::string& name) const {
  auto it = name_to_function_.find(name);  // reads take no lock (registry.cc:60-66)
  if (it == name_to_function_.end()) return Status::KeyError("No function registered with name: " + name);
  return it->second;
}
std::vector<std::string> FunctionRegistry::GetFunctionNames() const {
  std::vector<std::string> names;
  for (const auto& kv : name_to_function_) names.push_back(kv.first);
  std::sort(names.begin(), names.end());
  return names;
}
int FunctionRegistry::num_functions() const { return static_cast<int>(name_to_function_.size()); }

FunctionRegistry* GetFunctionRegistry() {  // registry.cc:115-150, hot-path subset
  static std::unique_ptr<FunctionRegistry> g_registry = [] {
    auto r = FunctionRegistry::Make();
    internal::RegisterScalarArithmetic(r.get());
    internal::RegisterScalarComparison(r.get());
    internal::RegisterScalarBoolean(r.get());
    internal::RegisterScalarValidity(r.get());
    internal::RegisterScalarFillNull(r.get());
    internal::RegisterScalarCast(r.get());
    internal::RegisterScalarAggregateBasic(r.get());
    internal::RegisterVectorSelection(r.get());
    internal::RegisterVectorSort(r.get());
    internal::RegisterHashAggregateBasic(r.get());
    internal::RegisterVectorHash(r.get());
    return r;
  }();
  return g_registry.get();
}

// ---- CallFunction (exec.cc:972-990) ---------------------------------------------------------------------
Result<Datum> CallFunction(const std::string& func_name, const std::vector<Datum>& args, const FunctionOptions* options,
                           ExecContext* ctx) {
  if (ctx == nullptr) return Status::Invalid("CallFunction: this backend needs an ExecContext bound to a device MemoryPool");
  AMD_ASSIGN_OR_RAISE(std::shared_ptr<Function> func, ctx->func_registry()->GetFunction(func_name));
  return func->Execute(args, options, ctx);
}
Result<Datum> CallFunction(const std::string& func_name, const std::vector<Datum>& args, ExecContext* ctx) {
  return CallFunction(func_name, args, nullptr, ctx);
}

Result<Datum> Filter(const Datum& values, const Datum& filter, const FilterOptions& options, ExecContext* ctx) {
  return CallFunction("filter", {values, filter}, &options, ctx);  // api_vector.cc:95-99
}
Result<Datum> Take(const Datum& values, const Datum& indices, const TakeOptions& options, ExecContext* ctx) {
  return CallFunction("take", {values, indices}, &options, ctx);  // api_vector.cc:101-107
}
Result<Datum> SortIndices(const Datum& values, SortOrder order, ExecContext* ctx) {
  SortOptions options({SortKey("not-used", order)});  // api_vector.cc:54-58
  return CallFunction("sort_indices", {values}, &options, ctx);
}
Result<Datum> SortIndices(const Datum& values, const SortOptions& options, ExecContext* ctx) {
  return CallFunction("sort_indices", {values}, &options, ctx);  // api_vector.cc:93-96
}
Result<Datum> NthToIndices(const Datum& values, int64_t n, ExecContext* ctx) {
  PartitionNthOptions options(n);  // api_vector.cc:43-48
  return CallFunction("partition_nth_indices", {values}, &options, ctx);
}
Result<Datum> Unique(const Datum& values, ExecContext* ctx) { return CallFunction("unique", {values}, nullptr, ctx); }
Result<Datum> ValueCounts(const Datum& values, ExecContext* ctx) { return CallFunction("value_counts", {values}, nullptr, ctx); }
Result<Datum> DictionaryEncode(const Datum& values, const DictionaryEncodeOptions& options, ExecContext* ctx) {
  return CallFunction("dictionary_encode", {values}, &options, ctx);
}
Result<Datum> Compare(const Datum& left, const Datum& right, CompareOperator op, ExecContext* ctx) {
  static const char* names[] = {"equal", "not_equal", "greater", "greater_equal", "less", "less_equal"};  // api_scalar.cc:122-146
  return CallFunction(names[op], {left, right}, nullptr, ctx);
}
static Result<Datum> Arith(const char* name, const Datum& l, const Datum& r, ArithmeticOptions o, ExecContext* ctx) {
  return CallFunction(o.check_overflow ? std::string(name) + "_checked" : std::string(name), {l, r}, nullptr, ctx);  // api_scalar.cc:52-63
}
Result<Datum> Add(const Datum& l, const Datum& r, ArithmeticOptions o, ExecContext* ctx) { return Arith("add", l, r, o, ctx); }
Result<Datum> Subtract(const Datum& l, const Datum& r, ArithmeticOptions o, ExecContext* ctx) { return Arith("subtract", l, r, o, ctx); }
Result<Datum> Multiply(const Datum& l, const Datum& r, ArithmeticOptions o, ExecContext* ctx) { return Arith("multiply", l, r, o, ctx); }
Result<Datum> Divide(const Datum& l, const Datum& r, ArithmeticOptions o, ExecContext* ctx) { return Arith("divide", l, r, o, ctx); }

}  // namespace compute
}  // namespace amdarrow
