// amdarrow_compute.h -- C++ host mirror of the reference's operator API for the hot path.
//
// Same names, argument meaning and error behaviour as arrow::compute
// (cpp/src/arrow/compute/{exec,function,kernel,registry,api_vector,api_scalar,api_aggregate}.h of
// the reference), re-stated in namespace `amdarrow` so that it can live next to a real Arrow
// build.  Every registered kernel's exec is a thin call into the C ABI (include/amdarrow.h);
// there are no CPU kernels here.  What a maintainer of the reference does instead of using
// this mirror -- registering the same C-ABI calls as arrow::compute kernels in an
// arrow::compute::FunctionRegistry -- is shown in INTEGRATION.md.
//
// Data model (consumed, not rebuilt -- SURVEY.md 2d): ArrayData{type, length, null_count,
// offset, buffers[0]=validity, buffers[1]=values}; Buffers here are HBM allocations.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "amdarrow.h"

namespace amdarrow {

// ---- Status / Result (arrow/status.h, arrow/result.h) ----------------------------------------
enum class StatusCode : char {
  OK = 0, OutOfMemory = 1, KeyError = 2, TypeError = 3, Invalid = 4, IOError = 5, CapacityError = 6,
  IndexError = 7, Cancelled = 8, UnknownError = 9, NotImplemented = 10
};

class Status {
 public:
  Status() = default;
  Status(StatusCode code, std::string msg) : code_(code), msg_(std::move(msg)) {}
  static Status OK() { return Status(); }
  static Status Invalid(std::string m) { return Status(StatusCode::Invalid, std::move(m)); }
  static Status KeyError(std::string m) { return Status(StatusCode::KeyError, std::move(m)); }
  static Status TypeError(std::string m) { return Status(StatusCode::TypeError, std::move(m)); }
  static Status IndexError(std::string m) { return Status(StatusCode::IndexError, std::move(m)); }
  static Status NotImplemented(std::string m) { return Status(StatusCode::NotImplemented, std::move(m)); }
  static Status CapacityError(std::string m) { return Status(StatusCode::CapacityError, std::move(m)); }
  static Status OutOfMemory(std::string m) { return Status(StatusCode::OutOfMemory, std::move(m)); }
  // A non-zero code returned by the C ABI, with amdarrow_last_error() as message.
  static Status FromAbi(int code);
  bool ok() const { return code_ == StatusCode::OK; }
  StatusCode code() const { return code_; }
  const std::string& message() const { return msg_; }
  bool IsInvalid() const { return code_ == StatusCode::Invalid; }
  bool IsKeyError() const { return code_ == StatusCode::KeyError; }
  bool IsIndexError() const { return code_ == StatusCode::IndexError; }
  bool IsNotImplemented() const { return code_ == StatusCode::NotImplemented; }
  bool IsCapacityError() const { return code_ == StatusCode::CapacityError; }
  std::string ToString() const;

 private:
  StatusCode code_ = StatusCode::OK;
  std::string msg_;
};

template <typename T>
class Result {
 public:
  Result(T value) : value_(std::move(value)) {}           // NOLINT implicit, like arrow::Result
  Result(Status st) : status_(std::move(st)) {}           // NOLINT
  bool ok() const { return status_.ok(); }
  const Status& status() const { return status_; }
  T& ValueOrDie() { return value_; }
  const T& ValueOrDie() const { return value_; }
  T& operator*() { return value_; }
  const T& operator*() const { return value_; }
  T* operator->() { return &value_; }
  const T* operator->() const { return &value_; }
  T MoveValueUnsafe() { return std::move(value_); }

 private:
  Status status_;
  T value_{};
};

#define AMD_RETURN_NOT_OK(expr)            \
  do {                                     \
    ::amdarrow::Status _st = (expr);       \
    if (!_st.ok()) return _st;             \
  } while (0)
#define AMD_CONCAT_(a, b) a##b
#define AMD_CONCAT(a, b) AMD_CONCAT_(a, b)
#define AMD_ASSIGN_OR_RAISE_IMPL(tmp, lhs, rexpr) \
  auto tmp = (rexpr);                             \
  if (!tmp.ok()) return tmp.status();             \
  lhs = tmp.MoveValueUnsafe();
#define AMD_ASSIGN_OR_RAISE(lhs, rexpr) AMD_ASSIGN_OR_RAISE_IMPL(AMD_CONCAT(_res_, __LINE__), lhs, rexpr)

// ---- types (arrow/type_fwd.h:277-313; only the primitive types on the path) -------------------------
struct Type {
  enum type { NA = 0, BOOL = 1, UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64, HALF_FLOAT, FLOAT, DOUBLE,
              STRUCT,       // the {min, max} / {values, counts} results
              DICTIONARY,   // dictionary_encode's result: int32 indices + ArrayData::dictionary
              BINARY, STRING,   // 32-bit offsets; buffers = validity | offsets | data (Filter / Take only)
              LIST,             // list<...> with int32 offsets: Grouper::MakeGroupings / ApplyGroupings; buffers = validity | offsets, child_data[0] = the values
              // the other value types array_take / array_filter accept (vector_selection.cc:2133-2176)
              LARGE_BINARY, LARGE_STRING,   // 64-bit offsets
              FIXED_SIZE_BINARY, DECIMAL128, DECIMAL256,  // buffers = validity | byte_width bytes per slot
              LARGE_LIST, FIXED_SIZE_LIST, MAP,  // MAP: a list of struct<key, item> (32-bit offsets)
              EXTENSION,                    // storage_type() carries the layout
              MAX_ID };
};
class DataType {
 public:
  explicit DataType(Type::type id) : id_(id) {}
  Type::type id() const { return id_; }
  int bit_width() const;
  std::string ToString() const;
  bool Equals(const DataType& o) const { return id_ == o.id_; }
  // DictionaryType::value_type (type.h:1360-1400); null for the other types and for dictionary_()
  const std::shared_ptr<DataType>& value_type() const { return value_type_; }
  void set_value_type(std::shared_ptr<DataType> t) { value_type_ = std::move(t); }
  // DictionaryType::index_type (int32 unless said otherwise); FixedSizeBinaryType::byte_width / Decimal widths;
  // FixedSizeListType::list_size; StructType fields (child i of the ArrayData); ExtensionType::storage_type + name
  const std::shared_ptr<DataType>& index_type() const { return index_type_; }
  void set_index_type(std::shared_ptr<DataType> t) { index_type_ = std::move(t); }
  int byte_width() const { return byte_width_; }
  void set_byte_width(int w) { byte_width_ = w; }
  int list_size() const { return list_size_; }
  void set_list_size(int n) { list_size_ = n; }
  const std::vector<std::pair<std::string, std::shared_ptr<DataType>>>& fields() const { return fields_; }
  void set_fields(std::vector<std::pair<std::string, std::shared_ptr<DataType>>> f) { fields_ = std::move(f); }
  const std::shared_ptr<DataType>& storage_type() const { return value_type_; }  // EXTENSION
  const std::string& extension_name() const { return name_; }
  void set_extension_name(std::string n) { name_ = std::move(n); }

 private:
  Type::type id_;
  std::shared_ptr<DataType> value_type_, index_type_;
  int byte_width_ = 0, list_size_ = 0;
  std::vector<std::pair<std::string, std::shared_ptr<DataType>>> fields_;
  std::string name_;
};
std::shared_ptr<DataType> null();
std::shared_ptr<DataType> boolean();
std::shared_ptr<DataType> uint8();
std::shared_ptr<DataType> int8();
std::shared_ptr<DataType> uint16();
std::shared_ptr<DataType> int16();
std::shared_ptr<DataType> uint32();
std::shared_ptr<DataType> int32();
std::shared_ptr<DataType> uint64();
std::shared_ptr<DataType> int64();
std::shared_ptr<DataType> float32();
std::shared_ptr<DataType> float64();
std::shared_ptr<DataType> struct_();  // field list carried by the StructScalar itself
std::shared_ptr<DataType> dictionary_();  // dictionary(int32(), <type of ArrayData::dictionary>)
std::shared_ptr<DataType> dictionary(std::shared_ptr<DataType> value_type);  // dictionary(int32(), value_type), type.h:1360
std::shared_ptr<DataType> list_();  // list(<type of child_data[0]>)
std::shared_ptr<DataType> binary();
std::shared_ptr<DataType> utf8();
inline bool is_binary_like(Type::type id) { return id == Type::BINARY || id == Type::STRING; }  // type_traits.h
inline bool is_large_binary_like(Type::type id) { return id == Type::LARGE_BINARY || id == Type::LARGE_STRING; }
std::shared_ptr<DataType> large_binary();
std::shared_ptr<DataType> large_utf8();
std::shared_ptr<DataType> fixed_size_binary(int byte_width);
std::shared_ptr<DataType> decimal128(int precision, int scale);  // (precision / scale ride along in the name only)
std::shared_ptr<DataType> decimal256(int precision, int scale);
std::shared_ptr<DataType> list(std::shared_ptr<DataType> value_type);
std::shared_ptr<DataType> large_list(std::shared_ptr<DataType> value_type);
std::shared_ptr<DataType> fixed_size_list(std::shared_ptr<DataType> value_type, int list_size);
std::shared_ptr<DataType> map(std::shared_ptr<DataType> key_type, std::shared_ptr<DataType> item_type);
std::shared_ptr<DataType> struct_(std::vector<std::pair<std::string, std::shared_ptr<DataType>>> fields);
std::shared_ptr<DataType> dictionary(std::shared_ptr<DataType> index_type, std::shared_ptr<DataType> value_type);
std::shared_ptr<DataType> extension(std::string name, std::shared_ptr<DataType> storage_type);
std::shared_ptr<DataType> TypeFromId(Type::type id);
bool is_integer(Type::type id);
bool is_signed_integer(Type::type id);
bool is_floating(Type::type id);

// ---- device memory (arrow/memory_pool.h:38-105, arrow/buffer.h:50, device.h:40-151) ---------------
// MemoryPool over HBM with the reference's two counters.  One pool = one amdarrow_ctx = one device
// and one stream.
class MemoryPool {
 public:
  static Result<std::shared_ptr<MemoryPool>> Make(int device = 0, void* stream = nullptr);
  ~MemoryPool();
  amdarrow_ctx* ctx() const { return ctx_; }
  int64_t bytes_allocated() const;
  int64_t max_memory() const;
  // MemoryPool::ReleaseUnused (memory_pool.h:92-96): hand the freed blocks the pool keeps for reuse back to the device
  void ReleaseUnused();
  int64_t cached_bytes() const;

 private:
  amdarrow_ctx* ctx_ = nullptr;
};

class Buffer {
 public:
  Buffer(void* data, int64_t size, std::shared_ptr<MemoryPool> pool, bool owned)
      : data_(data), size_(size), pool_(std::move(pool)), owned_(owned) {}
  ~Buffer();
  Buffer(const Buffer&) = delete;
  const uint8_t* data() const { return static_cast<const uint8_t*>(data_); }
  uint8_t* mutable_data() { return static_cast<uint8_t*>(data_); }
  int64_t size() const { return size_; }
  bool is_cpu() const { return pool_ == nullptr; }  // Buffer::is_cpu(), buffer.h:226
  // Wraps host memory (no ownership): only for the plumbing tests that register host-side toy kernels.
  static std::shared_ptr<Buffer> WrapHost(const void* data, int64_t size);

 private:
  void* data_;
  int64_t size_;
  std::shared_ptr<MemoryPool> pool_;
  bool owned_;
};
Result<std::shared_ptr<Buffer>> AllocateBuffer(int64_t nbytes, const std::shared_ptr<MemoryPool>& pool);
// Zero-filled bitmap (KernelContext::AllocateBitmap, kernel.cc:48-60)
Result<std::shared_ptr<Buffer>> AllocateBitmap(int64_t nbits, const std::shared_ptr<MemoryPool>& pool);

// ---- ArrayData / Scalar / ChunkedArray / Datum (array/data.h:73-245, scalar.h, datum.h:106) -------
constexpr int64_t kUnknownNullCount = -1;

struct ArrayData {
  std::shared_ptr<DataType> type;
  int64_t length = 0;
  int64_t null_count = 0;
  int64_t offset = 0;
  std::vector<std::shared_ptr<Buffer>> buffers;  // [0] validity (may be null), [1] values
  // StructArray children (array/data.h:232): only hash_min_max's {min, max} output has any
  std::vector<std::shared_ptr<ArrayData>> child_data;
  std::vector<std::string> child_names;
  std::shared_ptr<ArrayData> dictionary;  // DictionaryArray (array/data.h:240): buffers = the int32 indices

  static std::shared_ptr<ArrayData> Make(std::shared_ptr<DataType> type, int64_t length,
                                         std::vector<std::shared_ptr<Buffer>> buffers, int64_t null_count = kUnknownNullCount,
                                         int64_t offset = 0);
  std::shared_ptr<ArrayData> Slice(int64_t off, int64_t len) const;  // array/data.cc Slice
  bool MayHaveNulls() const { return null_count != 0 && buffers[0] != nullptr; }
  // The C-ABI view (= PrimitiveArg, kernels/util_internal.h:35-47)
  amdarrow_array abi() const;
  amdarrow_binary_array binary_abi() const;  // BINARY / STRING: buffers[1] offsets, buffers[2] data
  amdarrow_var_array var_abi() const;        // any offsets-based layout: (LARGE_)BINARY / STRING, (LARGE_)LIST, MAP
};

// Host -> device construction helpers (what a caller with host Arrow arrays does first).
Result<std::shared_ptr<ArrayData>> ArrayFromHost(const std::shared_ptr<MemoryPool>& pool, std::shared_ptr<DataType> type,
                                                 const void* values, int64_t length, const uint8_t* validity_bits = nullptr,
                                                 int64_t null_count = kUnknownNullCount);
// A StringArray / BinaryArray from host strings; `valid[i] == false` makes slot i null (its bytes are ignored).
Result<std::shared_ptr<ArrayData>> BinaryFromHost(const std::shared_ptr<MemoryPool>& pool, std::shared_ptr<DataType> type,
                                                  const std::vector<std::string>& items, const std::vector<bool>& valid = {});
// ... and back: (value, is_valid) per slot.
Result<std::vector<std::pair<std::string, bool>>> BinaryToHost(const ArrayData& a, const std::shared_ptr<MemoryPool>& pool);
// Device -> host copies for inspection.
Status CopyValuesToHost(const ArrayData& a, const std::shared_ptr<MemoryPool>& pool, void* out_values,
                        uint8_t* out_validity_bits);

struct Scalar {
  std::shared_ptr<DataType> type;
  bool is_valid = true;
  uint64_t bits = 0;  // raw value bits (little endian, low bytes)
  // StructScalar (scalar.h): min_max returns {min, max}
  std::vector<std::shared_ptr<Scalar>> fields;
  std::vector<std::string> field_names;
  template <typename T>
  static std::shared_ptr<Scalar> Make(std::shared_ptr<DataType> t, T v) {
    auto s = std::make_shared<Scalar>();
    s->type = std::move(t);
    static_assert(sizeof(T) <= 8, "primitive scalars only");
    __builtin_memcpy(&s->bits, &v, sizeof(T));
    return s;
  }
  static std::shared_ptr<Scalar> MakeNull(std::shared_ptr<DataType> t) {
    auto s = std::make_shared<Scalar>();
    s->type = std::move(t);
    s->is_valid = false;
    return s;
  }
};

class ChunkedArray {
 public:
  ChunkedArray(std::vector<std::shared_ptr<ArrayData>> chunks, std::shared_ptr<DataType> type);
  int num_chunks() const { return static_cast<int>(chunks_.size()); }
  const std::shared_ptr<ArrayData>& chunk(int i) const { return chunks_[i]; }
  const std::vector<std::shared_ptr<ArrayData>>& chunks() const { return chunks_; }
  int64_t length() const { return length_; }
  int64_t null_count() const { return null_count_; }
  const std::shared_ptr<DataType>& type() const { return type_; }

 private:
  std::vector<std::shared_ptr<ArrayData>> chunks_;
  std::shared_ptr<DataType> type_;
  int64_t length_ = 0, null_count_ = 0;
};

// arrow::Concatenate (array/concatenate.cc:428-449) for the path's fixed-width / boolean arrays
class MemoryPool;
Result<std::shared_ptr<ArrayData>> Concatenate(const std::vector<std::shared_ptr<ArrayData>>& arrays,
                                               const std::shared_ptr<MemoryPool>& pool);

// ---- Field / Schema / RecordBatch / Table (type.h:300-420, 1600-1700; record_batch.h:43-180;
// table.h:44-200): the members the path's meta functions use.  Columns live in HBM. -----------------
struct Field {
  std::string name;
  std::shared_ptr<DataType> type;
};
inline Field field(std::string name, std::shared_ptr<DataType> type) { return {std::move(name), std::move(type)}; }
class Schema {
 public:
  explicit Schema(std::vector<Field> fields) : fields_(std::move(fields)) {}
  int num_fields() const { return static_cast<int>(fields_.size()); }
  const Field& field(int i) const { return fields_[i]; }
  int GetFieldIndex(const std::string& name) const {  // -1 when absent (type.h:1648)
    for (int i = 0; i < num_fields(); ++i) if (fields_[i].name == name) return i;
    return -1;
  }

 private:
  std::vector<Field> fields_;
};
inline std::shared_ptr<Schema> schema(std::vector<Field> fields) { return std::make_shared<Schema>(std::move(fields)); }
class RecordBatch {
 public:
  static std::shared_ptr<RecordBatch> Make(std::shared_ptr<Schema> schema, int64_t num_rows,
                                           std::vector<std::shared_ptr<ArrayData>> columns) {
    return std::shared_ptr<RecordBatch>(new RecordBatch(std::move(schema), num_rows, std::move(columns)));
  }
  const std::shared_ptr<Schema>& schema() const { return schema_; }
  int64_t num_rows() const { return num_rows_; }
  int num_columns() const { return static_cast<int>(columns_.size()); }
  const std::shared_ptr<ArrayData>& column(int i) const { return columns_[i]; }
  std::shared_ptr<ArrayData> GetColumnByName(const std::string& name) const {  // record_batch.cc:164-167
    const int i = schema_->GetFieldIndex(name);
    return i < 0 ? nullptr : columns_[i];
  }

 private:
  RecordBatch(std::shared_ptr<Schema> s, int64_t n, std::vector<std::shared_ptr<ArrayData>> c)
      : schema_(std::move(s)), num_rows_(n), columns_(std::move(c)) {}
  std::shared_ptr<Schema> schema_;
  int64_t num_rows_;
  std::vector<std::shared_ptr<ArrayData>> columns_;
};
class Table {
 public:
  // num_rows = -1: taken from the first column, 0 without columns (table.cc:58-70)
  static std::shared_ptr<Table> Make(std::shared_ptr<Schema> schema, std::vector<std::shared_ptr<ChunkedArray>> columns,
                                     int64_t num_rows = -1) {
    if (num_rows < 0) num_rows = columns.empty() ? 0 : columns[0]->length();
    return std::shared_ptr<Table>(new Table(std::move(schema), num_rows, std::move(columns)));
  }
  const std::shared_ptr<Schema>& schema() const { return schema_; }
  int64_t num_rows() const { return num_rows_; }
  int num_columns() const { return static_cast<int>(columns_.size()); }
  const std::shared_ptr<ChunkedArray>& column(int i) const { return columns_[i]; }
  const std::vector<std::shared_ptr<ChunkedArray>>& columns() const { return columns_; }
  std::shared_ptr<ChunkedArray> GetColumnByName(const std::string& name) const {  // table.h:100-103
    const int i = schema_->GetFieldIndex(name);
    return i < 0 ? nullptr : columns_[i];
  }

 private:
  Table(std::shared_ptr<Schema> s, int64_t n, std::vector<std::shared_ptr<ChunkedArray>> c)
      : schema_(std::move(s)), num_rows_(n), columns_(std::move(c)) {}
  std::shared_ptr<Schema> schema_;
  int64_t num_rows_;
  std::vector<std::shared_ptr<ChunkedArray>> columns_;
};

struct ValueDescr {
  enum Shape { ANY, ARRAY, SCALAR };
  std::shared_ptr<DataType> type;
  Shape shape = ANY;
  static ValueDescr Array(std::shared_ptr<DataType> t) { return {std::move(t), ARRAY}; }
  static ValueDescr Scalar(std::shared_ptr<DataType> t) { return {std::move(t), SCALAR}; }
  std::string ToString() const;
};

class Datum {
 public:
  enum Kind { NONE, SCALAR, ARRAY, CHUNKED_ARRAY, RECORD_BATCH, TABLE, COLLECTION };  // datum.h:106
  Datum() = default;
  Datum(std::shared_ptr<ArrayData> a) : kind_(ARRAY), array_(std::move(a)) {}               // NOLINT
  Datum(std::shared_ptr<ChunkedArray> c) : kind_(CHUNKED_ARRAY), chunked_(std::move(c)) {}  // NOLINT
  Datum(std::shared_ptr<Scalar> s) : kind_(SCALAR), scalar_(std::move(s)) {}                // NOLINT
  Datum(std::shared_ptr<RecordBatch> b) : kind_(RECORD_BATCH), batch_(std::move(b)) {}      // NOLINT
  Datum(std::shared_ptr<Table> t) : kind_(TABLE), table_(std::move(t)) {}                   // NOLINT
  Kind kind() const { return kind_; }
  bool is_array() const { return kind_ == ARRAY; }
  bool is_scalar() const { return kind_ == SCALAR; }
  bool is_arraylike() const { return kind_ == ARRAY || kind_ == CHUNKED_ARRAY; }
  const std::shared_ptr<ArrayData>& array() const { return array_; }
  const std::shared_ptr<ChunkedArray>& chunked_array() const { return chunked_; }
  const std::shared_ptr<Scalar>& scalar() const { return scalar_; }
  const std::shared_ptr<RecordBatch>& record_batch() const { return batch_; }
  const std::shared_ptr<Table>& table() const { return table_; }
  std::shared_ptr<DataType> type() const;
  int64_t length() const;
  ValueDescr descr() const;

 private:
  Kind kind_ = NONE;
  std::shared_ptr<ArrayData> array_;
  std::shared_ptr<ChunkedArray> chunked_;
  std::shared_ptr<Scalar> scalar_;
  std::shared_ptr<RecordBatch> batch_;
  std::shared_ptr<Table> table_;
};

namespace compute {

// ---- options (api_vector.h:35-110, api_scalar.h:124-137, api_aggregate.h:46-59) ---------------------
struct FunctionOptions {
  virtual ~FunctionOptions() = default;
};
struct FilterOptions : public FunctionOptions {
  enum NullSelectionBehavior { DROP, EMIT_NULL };
  explicit FilterOptions(NullSelectionBehavior n = DROP) : null_selection_behavior(n) {}
  static FilterOptions Defaults() { return FilterOptions(); }
  NullSelectionBehavior null_selection_behavior = DROP;
};
struct TakeOptions : public FunctionOptions {
  explicit TakeOptions(bool boundscheck = true) : boundscheck(boundscheck) {}
  static TakeOptions Defaults() { return TakeOptions(true); }
  static TakeOptions NoBoundsCheck() { return TakeOptions(false); }
  bool boundscheck = true;
};
struct CastOptions : public FunctionOptions {  // cast.h:41-70 (the flags the number casts read)
  static CastOptions Safe(std::shared_ptr<DataType> to_type = nullptr) {
    CastOptions o;
    o.to_type = std::move(to_type);
    return o;
  }
  static CastOptions Unsafe(std::shared_ptr<DataType> to_type = nullptr) {
    CastOptions o;
    o.to_type = std::move(to_type);
    o.allow_int_overflow = o.allow_float_truncate = true;
    return o;
  }
  std::shared_ptr<DataType> to_type;
  bool allow_int_overflow = false, allow_float_truncate = false;
};
enum class SortOrder { Ascending, Descending };
struct ArraySortOptions : public FunctionOptions {
  explicit ArraySortOptions(SortOrder order = SortOrder::Ascending) : order(order) {}
  static ArraySortOptions Defaults() { return ArraySortOptions(); }
  SortOrder order;
};
struct DictionaryEncodeOptions : public FunctionOptions {  // api_vector.h:67-82
  enum NullEncodingBehavior { ENCODE, MASK };
  explicit DictionaryEncodeOptions(NullEncodingBehavior n = MASK) : null_encoding_behavior(n) {}
  static DictionaryEncodeOptions Defaults() { return DictionaryEncodeOptions(); }
  NullEncodingBehavior null_encoding_behavior = MASK;
};
struct PartitionNthOptions : public FunctionOptions {  // api_vector.h:80-88
  explicit PartitionNthOptions(int64_t pivot) : pivot(pivot) {}
  int64_t pivot;
};
struct SortKey {
  explicit SortKey(std::string name, SortOrder order = SortOrder::Ascending) : name(std::move(name)), order(order) {}
  std::string name;
  SortOrder order;
};
struct SortOptions : public FunctionOptions {
  explicit SortOptions(std::vector<SortKey> sort_keys = {}) : sort_keys(std::move(sort_keys)) {}
  static SortOptions Defaults() { return SortOptions(); }
  std::vector<SortKey> sort_keys;
};
struct MinMaxOptions : public FunctionOptions {  // api_aggregate.h:78-90
  enum Mode { SKIP = 0, EMIT_NULL };
  explicit MinMaxOptions(Mode m = SKIP) : null_handling(m) {}
  static MinMaxOptions Defaults() { return MinMaxOptions(); }
  Mode null_handling;
};
struct CountOptions : public FunctionOptions {
  enum Mode { COUNT_NON_NULL = 0, COUNT_NULL };
  explicit CountOptions(Mode m = COUNT_NON_NULL) : count_mode(m) {}
  static CountOptions Defaults() { return CountOptions(); }
  Mode count_mode;
};
enum CompareOperator : int8_t { EQUAL, NOT_EQUAL, GREATER, GREATER_EQUAL, LESS, LESS_EQUAL };
struct ArithmeticOptions : public FunctionOptions {
  ArithmeticOptions(bool check_overflow = false) : check_overflow(check_overflow) {}  // NOLINT
  bool check_overflow;
};

class FunctionRegistry;

// ---- ExecContext / ExecBatch (exec.h:58-218) ---------------------------------------------------------
constexpr int64_t kDefaultExecChunksize = std::numeric_limits<int64_t>::max();

class ExecContext {
 public:
  // `pool` owns the device context; `func_registry == nullptr` selects GetFunctionRegistry().
  explicit ExecContext(std::shared_ptr<MemoryPool> pool, FunctionRegistry* func_registry = nullptr);
  const std::shared_ptr<MemoryPool>& memory_pool() const { return pool_; }
  FunctionRegistry* func_registry() const { return func_registry_; }
  void set_exec_chunksize(int64_t n) { exec_chunksize_ = n; }
  int64_t exec_chunksize() const { return exec_chunksize_; }
  void set_preallocate_contiguous(bool v) { preallocate_contiguous_ = v; }
  bool preallocate_contiguous() const { return preallocate_contiguous_; }
  void set_use_threads(bool v) { use_threads_ = v; }  // "not yet used" in the reference either (exec.h:86-92)
  bool use_threads() const { return use_threads_; }

 private:
  std::shared_ptr<MemoryPool> pool_;
  FunctionRegistry* func_registry_;
  int64_t exec_chunksize_ = kDefaultExecChunksize;
  bool preallocate_contiguous_ = true;
  bool use_threads_ = true;
};

struct ExecBatch {
  ExecBatch() = default;
  ExecBatch(std::vector<Datum> values, int64_t length) : values(std::move(values)), length(length) {}
  // ExecBatch::Make (exec.cc:60-84): all array arguments must have equal length
  static Result<ExecBatch> Make(std::vector<Datum> values);
  std::vector<Datum> values;
  int64_t length = 0;
  const Datum& operator[](int i) const { return values[i]; }
  int num_values() const { return static_cast<int>(values.size()); }
  std::vector<ValueDescr> GetDescriptors() const;
};

namespace detail {
// ExecBatchIterator (exec.cc:158-252): equal-length contiguous batches <= max_chunksize, aligned
// to chunk boundaries of every chunked argument.
class ExecBatchIterator {
 public:
  static Result<std::unique_ptr<ExecBatchIterator>> Make(std::vector<Datum> args, int64_t max_chunksize);
  bool Next(ExecBatch* batch);
  int64_t length() const { return length_; }
  int64_t position() const { return position_; }

 private:
  std::vector<Datum> args_;
  std::vector<int> chunk_indexes_;
  std::vector<int64_t> chunk_positions_;
  int64_t position_ = 0, length_ = 0, max_chunksize_ = 0;
};
}  // namespace detail

// ---- kernels (kernel.h:94, 428-474, 504-721) ---------------------------------------------------------
class KernelContext {
 public:
  explicit KernelContext(ExecContext* ctx) : ctx_(ctx) {}
  ExecContext* exec_context() const { return ctx_; }
  const std::shared_ptr<MemoryPool>& memory_pool() const { return ctx_->memory_pool(); }
  amdarrow_ctx* abi() const { return ctx_->memory_pool() ? ctx_->memory_pool()->ctx() : nullptr; }
  Result<std::shared_ptr<Buffer>> Allocate(int64_t nbytes) { return AllocateBuffer(nbytes, ctx_->memory_pool()); }
  Result<std::shared_ptr<Buffer>> AllocateBitmap(int64_t nbits) { return ::amdarrow::AllocateBitmap(nbits, ctx_->memory_pool()); }
  void SetOptions(const FunctionOptions* o) { options_ = o; }
  const FunctionOptions* options() const { return options_; }

 private:
  ExecContext* ctx_;
  const FunctionOptions* options_ = nullptr;
};

struct NullHandling { enum type { INTERSECTION, COMPUTED_PREALLOCATE, COMPUTED_NO_PREALLOCATE, OUTPUT_NOT_NULL }; };
struct MemAllocation { enum type { PREALLOCATE, NO_PREALLOCATE }; };

// Matches a Type id exactly, or anything when `any` (match::Primitive() etc. reduce to a predicate).
struct InputType {
  std::function<bool(const ValueDescr&)> matches;
  std::string repr;
  static InputType Exact(Type::type id, ValueDescr::Shape shape = ValueDescr::ANY);
  static InputType Primitive();
  static InputType Integer();
  static InputType Any();
};
struct KernelSignature {
  std::vector<InputType> in_types;
  std::function<Result<std::shared_ptr<DataType>>(const std::vector<ValueDescr>&)> out_type;
  bool MatchesInputs(const std::vector<ValueDescr>& args) const;
};

using ArrayKernelExec = std::function<Status(KernelContext*, const ExecBatch&, Datum*)>;  // kernel.h:94

struct Kernel {
  KernelSignature signature;
};
struct ArrayKernel : Kernel {
  ArrayKernelExec exec;
  bool can_write_into_slices = true;
};
struct ScalarKernel : ArrayKernel {
  NullHandling::type null_handling = NullHandling::INTERSECTION;   // kernel.h:574-575
  MemAllocation::type mem_allocation = MemAllocation::PREALLOCATE;
};
struct VectorKernel : ArrayKernel {
  NullHandling::type null_handling = NullHandling::COMPUTED_NO_PREALLOCATE;  // kernel.h:617-618
  MemAllocation::type mem_allocation = MemAllocation::NO_PREALLOCATE;
  bool can_execute_chunkwise = true;
  bool output_chunked = true;
};
struct KernelState { virtual ~KernelState() = default; };
// ScalarAggregateKernel (kernel.h:634-679): one state per batch, merged into the first, finalized once
struct ScalarAggregateKernel : Kernel {
  std::function<Result<std::unique_ptr<KernelState>>(KernelContext*, const std::shared_ptr<DataType>&, const FunctionOptions*)> init;
  std::function<Status(KernelContext*, KernelState*, const ExecBatch&)> consume;
  std::function<Status(KernelContext*, KernelState&&, KernelState*)> merge;
  std::function<Status(KernelContext*, KernelState*, Datum*)> finalize;
};
struct HashAggregateKernel : Kernel {  // kernel.h:681-721 (first constructor's order: init, consume, merge, finalize)
  std::function<Result<std::unique_ptr<KernelState>>(KernelContext*, const std::shared_ptr<DataType>&, const FunctionOptions*)> init;
  std::function<Status(KernelContext*, KernelState*, const ExecBatch&)> consume;  // {argument, group ids, num_groups}
  std::function<Status(KernelContext*, KernelState&&, KernelState*)> merge;
  std::function<Status(KernelContext*, KernelState*, Datum*)> finalize;
};

// ---- functions (function.h:117-355) ----------------------------------------------------------------
struct Arity {
  int num_args;
  bool is_varargs = false;
  static Arity Unary() { return {1}; }
  static Arity Binary() { return {2}; }
  static Arity Ternary() { return {3}; }
  static Arity VarArgs(int min_args = 0) { return {min_args, true}; }
};
struct FunctionDoc {
  std::string summary, description;
  std::vector<std::string> arg_names;
  std::string options_class;
};

class Function {
 public:
  enum Kind { SCALAR, VECTOR, SCALAR_AGGREGATE, HASH_AGGREGATE, META };
  virtual ~Function() = default;
  const std::string& name() const { return name_; }
  Kind kind() const { return kind_; }
  const Arity& arity() const { return arity_; }
  const FunctionDoc& doc() const { return doc_; }
  const FunctionOptions* default_options() const { return default_options_; }
  virtual int num_kernels() const { return 0; }
  // function.cc:41-56: arity check -> Invalid
  Status CheckArity(int passed_num_args) const;
  // function.cc:157-201
  virtual Result<Datum> Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const;

 protected:
  Function(std::string name, Kind kind, Arity arity, FunctionDoc doc, const FunctionOptions* default_options)
      : name_(std::move(name)), kind_(kind), arity_(arity), doc_(std::move(doc)), default_options_(default_options) {}
  std::string name_;
  Kind kind_;
  Arity arity_;
  FunctionDoc doc_;
  const FunctionOptions* default_options_;
};

class ScalarFunction : public Function {
 public:
  ScalarFunction(std::string name, Arity arity, FunctionDoc doc = {}, const FunctionOptions* default_options = nullptr)
      : Function(std::move(name), SCALAR, arity, std::move(doc), default_options) {}
  Status AddKernel(ScalarKernel kernel);
  int num_kernels() const override { return static_cast<int>(kernels_.size()); }
  Result<const ScalarKernel*> DispatchExact(const std::vector<ValueDescr>& values) const;
  // function.h:186-193 / function.cc:152-155: the kernel for `values` after implicit conversions, which are written
  // back into `values` (Execute casts the arguments to them, function.cc:175-176).  Default: DispatchExact.
  virtual Result<const ScalarKernel*> DispatchBest(std::vector<ValueDescr>* values) const { return DispatchExact(*values); }
  Result<Datum> Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override;

 private:
  std::vector<ScalarKernel> kernels_;
};
class VectorFunction : public Function {
 public:
  VectorFunction(std::string name, Arity arity, FunctionDoc doc = {}, const FunctionOptions* default_options = nullptr)
      : Function(std::move(name), VECTOR, arity, std::move(doc), default_options) {}
  Status AddKernel(VectorKernel kernel);
  int num_kernels() const override { return static_cast<int>(kernels_.size()); }
  Result<const VectorKernel*> DispatchExact(const std::vector<ValueDescr>& values) const;
  Result<Datum> Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override;

 private:
  std::vector<VectorKernel> kernels_;
};
class ScalarAggregateFunction : public Function {
 public:
  ScalarAggregateFunction(std::string name, Arity arity, FunctionDoc doc = {}, const FunctionOptions* default_options = nullptr)
      : Function(std::move(name), SCALAR_AGGREGATE, arity, std::move(doc), default_options) {}
  Status AddKernel(ScalarAggregateKernel kernel);
  int num_kernels() const override { return static_cast<int>(kernels_.size()); }
  Result<const ScalarAggregateKernel*> DispatchExact(const std::vector<ValueDescr>& values) const;
  // ScalarAggExecutor (exec.cc:802-903): consume every batch into a fresh state, merge, finalize -> a scalar
  Result<Datum> Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override;

 private:
  std::vector<ScalarAggregateKernel> kernels_;
};
class HashAggregateFunction : public Function {
 public:
  HashAggregateFunction(std::string name, Arity arity, FunctionDoc doc = {}, const FunctionOptions* default_options = nullptr)
      : Function(std::move(name), HASH_AGGREGATE, arity, std::move(doc), default_options) {}
  Status AddKernel(HashAggregateKernel kernel);
  int num_kernels() const override { return static_cast<int>(kernels_.size()); }
  Result<const HashAggregateKernel*> DispatchExact(const std::vector<ValueDescr>& values) const;
  // function.cc:193-195: Status::NotImplemented("Direct execution of HASH_AGGREGATE functions")
  Result<Datum> Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override;

 private:
  std::vector<HashAggregateKernel> kernels_;
};
class MetaFunction : public Function {
 public:
  Result<Datum> Execute(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override;

 protected:
  MetaFunction(std::string name, Arity arity, FunctionDoc doc = {}, const FunctionOptions* default_options = nullptr)
      : Function(std::move(name), META, arity, std::move(doc), default_options) {}
  virtual Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const = 0;
};

// ---- registry (registry.h:45-84) ---------------------------------------------------------------------
class FunctionRegistry {
 public:
  static std::unique_ptr<FunctionRegistry> Make();
  Status AddFunction(std::shared_ptr<Function> function, bool allow_overwrite = false);
  Status AddAlias(const std::string& target_name, const std::string& source_name);
  Result<std::shared_ptr<Function>> GetFunction(const std::string& name) const;
  std::vector<std::string> GetFunctionNames() const;
  int num_functions() const;

 private:
  mutable std::mutex mutex_;  // taken on mutation only (registry.cc:35-58)
  std::unordered_map<std::string, std::shared_ptr<Function>> name_to_function_;
};
// The default registry of this backend: every function on the hot path, HIP-backed.
FunctionRegistry* GetFunctionRegistry();

namespace internal {
// arrow::internal::RechunkArraysConsistently (array/util.cc:672-745): cut every group of chunks at
// the union of all groups' chunk boundaries; empty chunks disappear.  Zero-copy slices.
std::vector<std::vector<std::shared_ptr<ArrayData>>> RechunkArraysConsistently(
    const std::vector<std::vector<std::shared_ptr<ArrayData>>>& groups);
// GetTakeIndices (vector_selection.cc:92-237, api_vector.h:190): uint16 / uint32 positions of the
// selected slots; nullable under EMIT_NULL.
Result<std::shared_ptr<ArrayData>> GetTakeIndices(const ArrayData& filter, FilterOptions::NullSelectionBehavior null_selection,
                                                  ExecContext* ctx);
// The Register* hooks of the kernel files (registry_internal.h), HIP-backed.
void RegisterVectorSelection(FunctionRegistry* registry);  // array_filter, filter, array_take, take
void RegisterVectorSort(FunctionRegistry* registry);       // array_sort_indices, sort_indices, partition_nth_indices
void RegisterScalarComparison(FunctionRegistry* registry); // equal ... less_equal
void RegisterScalarArithmetic(FunctionRegistry* registry); // add ... divide(_checked)
void RegisterScalarBoolean(FunctionRegistry* registry);    // invert, and, and_not, or, xor, *_kleene
void RegisterScalarValidity(FunctionRegistry* registry);   // is_valid, is_null, is_nan
void RegisterScalarFillNull(FunctionRegistry* registry);   // fill_null
void RegisterScalarAggregateBasic(FunctionRegistry* registry);  // count, sum, mean, min_max
void RegisterHashAggregateBasic(FunctionRegistry* registry);  // hash_count, hash_sum, hash_mean, hash_min_max
void RegisterVectorHash(FunctionRegistry* registry);          // unique, value_counts, dictionary_encode

// Grouper / GroupBy (api_aggregate.h:313-403)
class Grouper {
 public:
  ~Grouper();
  static Result<std::unique_ptr<Grouper>> Make(const std::vector<ValueDescr>& descrs, ExecContext* ctx);
  Result<Datum> Consume(const ExecBatch& batch);  // uint32 group ids, first-appearance order
  Result<Datum> Lookup(const ExecBatch& batch);   // ids of rows whose keys are all known: no growth, KeyError otherwise
  Result<ExecBatch> GetUniques();
  uint32_t num_groups() const;
  // api_aggregate.h:360-377 / hash_aggregate.cc:979-1018: list<int32>[num_groups], element g = the rows whose id is g,
  // in row order.  ids: uint32 without nulls ("MakeGroupings with null ids" otherwise).
  static Result<std::shared_ptr<ArrayData>> MakeGroupings(const ArrayData& ids, uint32_t num_groups, ExecContext* ctx);
  // api_aggregate.h:379-395 / hash_aggregate.cc:968-977: list<array.type> with the groupings' offsets over
  // take(array, groupings' values)
  static Result<std::shared_ptr<ArrayData>> ApplyGroupings(const ArrayData& groupings, const ArrayData& array, ExecContext* ctx);

 private:
  Grouper() = default;
  // key column k as the fixed-width grouper sees it: variable-width columns by the dense ids of their own grouper,
  // dictionary columns by their indices (DictionaryKeyEncoder, hash_aggregate.cc:198-235)
  Result<std::vector<std::shared_ptr<ArrayData>>> FixedColumns(const ExecBatch& batch, bool insert);
  amdarrow_grouper* impl_ = nullptr;
  ExecContext* ctx_ = nullptr;
  std::vector<ValueDescr> descrs_;
  std::vector<amdarrow_binary_grouper*> var_;             // per key column, null unless (large_)binary / utf8
  std::vector<std::shared_ptr<ArrayData>> dictionaries_;  // per key column: the dictionary of the first batch
};
struct Aggregate {
  std::string function;
  const FunctionOptions* options;
};
// Returns the columns of the reference's StructArray{<agg fn name>..., key_0...} as a COLLECTION-like
// vector: one ArrayData per aggregate followed by one per key, plus their field names.
struct GroupByResult {
  std::vector<std::string> field_names;
  std::vector<std::shared_ptr<ArrayData>> columns;
  int64_t length = 0;
};
Result<GroupByResult> GroupBy(const std::vector<Datum>& arguments, const std::vector<Datum>& keys,
                              const std::vector<Aggregate>& aggregates, ExecContext* ctx);
}  // namespace internal

// ---- the entry point (exec.h:229-237) ----------------------------------------------------------------
Result<Datum> CallFunction(const std::string& func_name, const std::vector<Datum>& args, const FunctionOptions* options,
                           ExecContext* ctx);
Result<Datum> CallFunction(const std::string& func_name, const std::vector<Datum>& args, ExecContext* ctx);

// api_vector.h / api_scalar.h convenience wrappers (one-liners onto CallFunction, api_vector.cc:49-107)
Result<Datum> Filter(const Datum& values, const Datum& filter, const FilterOptions& options, ExecContext* ctx);
Result<Datum> Take(const Datum& values, const Datum& indices, const TakeOptions& options, ExecContext* ctx);
Result<Datum> SortIndices(const Datum& values, SortOrder order, ExecContext* ctx);
Result<Datum> SortIndices(const Datum& values, const SortOptions& options, ExecContext* ctx);  // api_vector.cc:93-96
Result<Datum> NthToIndices(const Datum& values, int64_t n, ExecContext* ctx);                  // api_vector.cc:43-48
Result<Datum> Unique(const Datum& values, ExecContext* ctx);                                   // api_vector.cc:115-118
Result<Datum> ValueCounts(const Datum& values, ExecContext* ctx);                              // api_vector.cc:130-133
Result<Datum> DictionaryEncode(const Datum& values, const DictionaryEncodeOptions& options, ExecContext* ctx);  // :120-124
// cast.h:128-160.  Numbers <-> numbers (not floating point -> integer), null -> numbers, dictionary -> its value type.
Result<Datum> Cast(const Datum& value, const CastOptions& options, ExecContext* ctx);
Result<Datum> Cast(const Datum& value, std::shared_ptr<DataType> to_type, const CastOptions& options, ExecContext* ctx);
namespace internal {
// codegen_internal.cc:190-268, the DispatchBest helpers
std::shared_ptr<DataType> CommonNumeric(const std::vector<ValueDescr>& descrs);
void EnsureDictionaryDecoded(std::vector<ValueDescr>* descrs);
void ReplaceNullWithOtherType(std::vector<ValueDescr>* descrs);
void ReplaceTypes(const std::shared_ptr<DataType>& type, std::vector<ValueDescr>* descrs);
// cast.cc:228-243: the arguments cast (safely) to the descriptors DispatchBest settled on
Result<std::vector<Datum>> Cast(std::vector<Datum> datums, const std::vector<ValueDescr>& descrs, ExecContext* ctx);
void RegisterScalarCast(FunctionRegistry* registry);  // "cast" (cast.cc:120-160)
}  // namespace internal
Result<Datum> Compare(const Datum& left, const Datum& right, CompareOperator op, ExecContext* ctx);
Result<Datum> Add(const Datum& left, const Datum& right, ArithmeticOptions options, ExecContext* ctx);
Result<Datum> Subtract(const Datum& left, const Datum& right, ArithmeticOptions options, ExecContext* ctx);
Result<Datum> Multiply(const Datum& left, const Datum& right, ArithmeticOptions options, ExecContext* ctx);
Result<Datum> Divide(const Datum& left, const Datum& right, ArithmeticOptions options, ExecContext* ctx);

}  // namespace compute
}  // namespace amdarrow
