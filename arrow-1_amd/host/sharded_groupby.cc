// sharded_groupby.cc -- see sharded_groupby.h.  Everything device-side goes through the C ABI (include/amdarrow.h).
#include "sharded_groupby.h"

#include <algorithm>
#include <cstring>

namespace amdarrow {
namespace compute {
namespace internal {

namespace {
constexpr int64_t kDenseLimit = int64_t(1) << 23;
constexpr int64_t kInf = int64_t(1) << 62;
int64_t ValueBytesOf(const DataType& t, int64_t n) { return t.bit_width() == 1 ? (n + 7) / 8 : n * t.bit_width() / 8; }
bool IsSignedId(Type::type id) { return id == Type::INT8 || id == Type::INT16 || id == Type::INT32 || id == Type::INT64; }
}  // namespace

Result<std::unique_ptr<ShardedGroupBy>> ShardedGroupBy::Make(ExecContext* ctx, std::shared_ptr<Collectives> coll, std::shared_ptr<DataType> key_type,
                                                             std::shared_ptr<DataType> value_type, int64_t group_capacity) {
  if (!ctx || !ctx->memory_pool()) return Status::Invalid("ShardedGroupBy: an ExecContext with a device MemoryPool is required");
  if (key_type->bit_width() < 8 || !is_integer(key_type->id())) return Status::NotImplemented("ShardedGroupBy: integer key columns");
  if (value_type->bit_width() != 64) return Status::NotImplemented("ShardedGroupBy: 64-bit value columns");
  auto g = std::unique_ptr<ShardedGroupBy>(new ShardedGroupBy());
  g->ctx_ = ctx;
  g->coll_ = std::move(coll);
  g->key_type_ = std::move(key_type);
  g->value_type_ = std::move(value_type);
  g->capacity_ = group_capacity;
  g->is_float_ = is_floating(g->value_type_->id());
  return g;
}

Result<ShardedGroupBy::Local> ShardedGroupBy::LocalGroupBy(const ArrayData* values, const ArrayData& keys, const ArrayData* mask) {
  KernelContext kctx(ctx_);
  Local t;
  AMD_ASSIGN_OR_RAISE(t.sums, kctx.Allocate(8 * capacity_));
  AMD_ASSIGN_OR_RAISE(t.counts, kctx.Allocate(8 * capacity_));
  AMD_ASSIGN_OR_RAISE(t.keys, kctx.Allocate(ValueBytesOf(*keys.type, capacity_) + 8));
  AMD_ASSIGN_OR_RAISE(t.key_validity, kctx.AllocateBitmap(capacity_ + 64));
  amdarrow_array v = values->abi(), k = keys.abi(), m{};
  if (mask) m = mask->abi();
  const int rc = amdarrow_filter_groupby(kctx.abi(), &v, &k, mask ? &m : nullptr, capacity_, t.sums->mutable_data(),
                                         reinterpret_cast<int64_t*>(t.counts->mutable_data()), t.keys->mutable_data(),
                                         t.key_validity->mutable_data(), nullptr, &t.num_groups);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return t;
}

// Global [lo, hi] of the non-null group keys so far on any rank: one 16-byte all-reduce(MIN) over (min, -max).
Status ShardedGroupBy::KeyRange(const Local& t, int64_t* lo, int64_t* hi) {
  amdarrow_ctx* abi = ctx_->memory_pool()->ctx();
  int64_t mm[2] = {kInf, kInf};
  if (t.num_groups > 0) {
    amdarrow_array a{};
    a.values = t.keys->data(); a.validity = t.key_validity->data(); a.length = t.num_groups; a.offset = 0;
    a.type_id = static_cast<int32_t>(key_type_->id()); a.null_count = -1;
    amdarrow_agg agg;
    const int rc = amdarrow_aggregate(abi, &a, &agg);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    if (agg.valid_count > 0) {
      if (!IsSignedId(key_type_->id()) && (agg.max >> 62) != 0) { mm[0] = -kInf; mm[1] = -kInf; }  // an unsigned key beyond int64: no dense plan
      else { mm[0] = static_cast<int64_t>(agg.min); mm[1] = -static_cast<int64_t>(agg.max); }
    }
  }
  AMD_ASSIGN_OR_RAISE(auto buf, AllocateBuffer(64, ctx_->memory_pool()));
  int rc = amdarrow_memcpy_h2d(abi, buf->mutable_data(), mm, 16);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  AMD_RETURN_NOT_OK(coll_->AllReduce(buf->mutable_data(), 2, Collectives::INT64, Collectives::MIN, amdarrow_ctx_stream(abi)));
  rc = amdarrow_memcpy_d2h(abi, mm, buf->data(), 16);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  if (mm[0] == kInf) { *lo = 0; *hi = -1; }
  else { *lo = mm[0]; *hi = -mm[1]; }
  return Status::OK();
}

Status ShardedGroupBy::Plan(const Local& t) {
  int64_t lo, hi;
  AMD_RETURN_NOT_OK(KeyRange(t, &lo, &hi));
  if (have_seen_ && hi >= lo) { lo = std::min(lo, seen_lo_); hi = std::max(hi, seen_hi_); }
  else if (have_seen_) { lo = seen_lo_; hi = seen_hi_; }
  if (hi >= lo) { have_seen_ = true; seen_lo_ = lo; seen_hi_ = hi; }
  const bool wide = lo <= -kInf / 2 || hi >= kInf / 2;
  const int64_t R = wide ? kInf : hi - lo + 1;
  if (!wide && R > 0 && R <= kDenseLimit) {
    const int64_t pad = std::max<int64_t>(1024, R / 8);  // head-room: later steps may see keys a little outside the range so far
    kmin_ = lo - pad;
    range_ = R + 2 * pad;
    AMD_ASSIGN_OR_RAISE(acc_, AllocateBuffer(8 * 2 * (range_ + 2), ctx_->memory_pool()));
    AMD_ASSIGN_OR_RAISE(owner_, AllocateBuffer(8 * (range_ + 2), ctx_->memory_pool()));
    plan_ = kDense;
  } else {
    plan_ = kGather;
  }
  return Status::OK();
}

GroupByResult ShardedGroupBy::Wrap(std::shared_ptr<Buffer> sums, std::shared_ptr<Buffer> counts, std::shared_ptr<Buffer> keys,
                                   std::shared_ptr<Buffer> key_validity, int64_t G, int64_t key_nulls) {
  GroupByResult r;
  r.length = G;
  r.field_names = {"hash_sum", "hash_count", "key_0"};
  // hash_sum is null where no valid value met the group (GroupedSumImpl::Finalize, hash_aggregate.cc:598-617): the host
  // mirror's GroupBy derives that bitmap from the counts; here the counts column travels next to it
  auto sum_type = is_float_ ? float64() : (IsSignedId(value_type_->id()) ? int64() : uint64());
  r.columns.push_back(ArrayData::Make(sum_type, G, {nullptr, std::move(sums)}, kUnknownNullCount));
  r.columns.push_back(ArrayData::Make(int64(), G, {nullptr, std::move(counts)}, 0));
  r.columns.push_back(ArrayData::Make(key_type_, G, {key_nulls != 0 ? std::move(key_validity) : nullptr, std::move(keys)}, key_nulls));
  return r;
}

Result<GroupByResult> ShardedGroupBy::DenseStep(const Local& t, bool* left_plan) {
  amdarrow_ctx* abi = ctx_->memory_pool()->ctx();
  void* stream = amdarrow_ctx_stream(abi);
  const int key_id = static_cast<int>(key_type_->id());
  int rc = amdarrow_dense_merge_scatter(abi, t.keys->data(), key_id, t.key_validity->data(), /*valid_is_bitmap=*/1, t.sums->data(),
                                        reinterpret_cast<const int64_t*>(t.counts->data()), t.num_groups, kmin_, range_, coll_->rank(),
                                        is_float_ ? 1 : 0, acc_->mutable_data(), owner_->mutable_data());
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  // sums, counts and the out-of-range count in ONE message; who saw each key first (and where in its order) in the other
  AMD_RETURN_NOT_OK(coll_->AllReduce(acc_->mutable_data(), 2 * (range_ + 2), is_float_ ? Collectives::FLOAT64 : Collectives::INT64, Collectives::SUM, stream));
  AMD_RETURN_NOT_OK(coll_->AllReduce(owner_->mutable_data(), range_ + 2, Collectives::INT64, Collectives::MIN, stream));
  const int64_t rows = range_ + 1;
  const auto& pool = ctx_->memory_pool();
  AMD_ASSIGN_OR_RAISE(auto out_keys, AllocateBuffer(ValueBytesOf(*key_type_, rows) + 8, pool));
  AMD_ASSIGN_OR_RAISE(auto out_valid, AllocateBuffer(rows + 8, pool));
  AMD_ASSIGN_OR_RAISE(auto out_sums, AllocateBuffer(8 * rows, pool));
  AMD_ASSIGN_OR_RAISE(auto out_counts, AllocateBuffer(8 * rows, pool));
  AMD_ASSIGN_OR_RAISE(auto word, AllocateBuffer(64, pool));
  rc = amdarrow_dense_merge_order(abi, acc_->data(), owner_->data(), kmin_, range_, coll_->world(), capacity_, key_id, is_float_ ? 1 : 0, rows,
                                  out_keys->mutable_data(), out_valid->mutable_data(), out_sums->mutable_data(),
                                  reinterpret_cast<int64_t*>(out_counts->mutable_data()), reinterpret_cast<int64_t*>(word->mutable_data()));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  int64_t G = 0;
  uint64_t bad_bits = 0;
  rc = amdarrow_memcpy_d2h(abi, &G, word->data(), 8);
  if (rc == AMDARROW_OK) rc = amdarrow_memcpy_d2h(abi, &bad_bits, acc_->data() + 16 * (range_ + 1), 8);  // acc[R + 1][0]
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  double bad_f;
  memcpy(&bad_f, &bad_bits, 8);
  *left_plan = is_float_ ? bad_f != 0.0 : bad_bits != 0;
  if (*left_plan) return GroupByResult{};
  // the validity came back as one byte per group: to a bitmap, and the null key's count
  AMD_ASSIGN_OR_RAISE(auto bitmap, AllocateBitmap(G + 64, pool));
  std::vector<uint8_t> bytes(static_cast<size_t>(G) + 8), bits(static_cast<size_t>((G + 7) / 8) + 8, 0);
  int64_t nulls = 0;
  if (G > 0) {
    rc = amdarrow_memcpy_d2h(abi, bytes.data(), out_valid->data(), G);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    for (int64_t i = 0; i < G; ++i) { if (bytes[i]) bits[i >> 3] |= uint8_t(1u << (i & 7)); else ++nulls; }
    rc = amdarrow_memcpy_h2d(abi, bitmap->mutable_data(), bits.data(), (G + 7) / 8);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  return Wrap(out_sums, out_counts, out_keys, bitmap, G, nulls);
}

Result<GroupByResult> ShardedGroupBy::GatherStep(const Local& t) {
  amdarrow_ctx* abi = ctx_->memory_pool()->ctx();
  void* stream = amdarrow_ctx_stream(abi);
  const auto& pool = ctx_->memory_pool();
  const int world = coll_->world();
  // the table sizes: 8 bytes per rank
  AMD_ASSIGN_OR_RAISE(auto gsend, AllocateBuffer(64, pool));
  AMD_ASSIGN_OR_RAISE(auto grecv, AllocateBuffer(8 * world + 64, pool));
  int rc = amdarrow_memcpy_h2d(abi, gsend->mutable_data(), &t.num_groups, 8);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  AMD_RETURN_NOT_OK(coll_->AllGather(gsend->data(), grecv->mutable_data(), 8, stream));
  std::vector<int64_t> g_all(world);
  rc = amdarrow_memcpy_d2h(abi, g_all.data(), grecv->data(), 8 * world);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  int64_t gmax = 0;
  for (int64_t g : g_all) gmax = std::max(gmax, g);
  gmax = (gmax + 63) / 64 * 64;  // whole bitmap words per rank: the gathered validity bitmaps line up
  if (gmax == 0) return Wrap(t.sums, t.counts, t.keys, t.key_validity, 0, 0);
  const int kw = key_type_->bit_width() / 8;
  const int64_t rows = gmax * world;
  // this rank's table is already padded: the local buffers hold `capacity_` rows (capacity_ >= gmax is required)
  if (gmax > capacity_) return Status::CapacityError("ShardedGroupBy: a rank holds more groups than group_capacity");
  AMD_ASSIGN_OR_RAISE(auto keys, AllocateBuffer(kw * rows + 8, pool));
  AMD_ASSIGN_OR_RAISE(auto valid, AllocateBitmap(rows + 64, pool));
  AMD_ASSIGN_OR_RAISE(auto sums, AllocateBuffer(8 * rows, pool));
  AMD_ASSIGN_OR_RAISE(auto counts, AllocateBuffer(8 * rows, pool));
  AMD_RETURN_NOT_OK(coll_->AllGather(t.keys->data(), keys->mutable_data(), kw * gmax, stream));
  AMD_RETURN_NOT_OK(coll_->AllGather(t.key_validity->data(), valid->mutable_data(), gmax / 8, stream));
  AMD_RETURN_NOT_OK(coll_->AllGather(t.sums->data(), sums->mutable_data(), 8 * gmax, stream));
  AMD_RETURN_NOT_OK(coll_->AllGather(t.counts->data(), counts->mutable_data(), 8 * gmax, stream));
  // which gathered rows are real: rank r's first g_all[r] of its gmax
  std::vector<uint8_t> live(static_cast<size_t>(rows / 8) + 8, 0);
  for (int r = 0; r < world; ++r)
    for (int64_t i = 0; i < g_all[r]; ++i) { const int64_t at = r * gmax + i; live[at >> 3] |= uint8_t(1u << (at & 7)); }
  AMD_ASSIGN_OR_RAISE(auto mask_bits, AllocateBitmap(rows + 64, pool));
  rc = amdarrow_memcpy_h2d(abi, mask_bits->mutable_data(), live.data(), rows / 8);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  // re-aggregate with the same fused kernel: rank-major concatenation keeps first-appearance order
  auto karr = ArrayData::Make(key_type_, rows, {valid, keys}, kUnknownNullCount);
  auto marr = ArrayData::Make(boolean(), rows, {nullptr, mask_bits}, 0);
  auto sum_type = is_float_ ? float64() : int64();
  auto sarr = ArrayData::Make(sum_type, rows, {nullptr, sums}, 0);
  auto carr = ArrayData::Make(int64(), rows, {nullptr, counts}, 0);
  AMD_ASSIGN_OR_RAISE(Local ms, LocalGroupBy(sarr.get(), *karr, marr.get()));   // sums of the partial sums
  AMD_ASSIGN_OR_RAISE(Local mc, LocalGroupBy(carr.get(), *karr, marr.get()));   // sums of the partial counts
  int64_t valid_keys = ms.num_groups;
  if (ms.num_groups > 0) {
    rc = amdarrow_count_set_bits(abi, ms.key_validity->data(), 0, ms.num_groups, &valid_keys);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  return Wrap(ms.sums, mc.sums, ms.keys, ms.key_validity, ms.num_groups, ms.num_groups - valid_keys);
}

Result<GroupByResult> ShardedGroupBy::Run(const Datum& values, const Datum& keys, const Datum& mask) {
  if (!values.is_array() || !keys.is_array()) return Status::NotImplemented("ShardedGroupBy: array arguments");
  const ArrayData* m = mask.is_array() ? mask.array().get() : nullptr;
  AMD_ASSIGN_OR_RAISE(Local t, LocalGroupBy(values.array().get(), *keys.array(), m));
  if (coll_->world() == 1 && !merge_alone_) {
    int64_t valid_keys = t.num_groups;
    if (t.num_groups > 0) {
      const int rc = amdarrow_count_set_bits(ctx_->memory_pool()->ctx(), t.key_validity->data(), 0, t.num_groups, &valid_keys);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    return Wrap(t.sums, t.counts, t.keys, t.key_validity, t.num_groups, t.num_groups - valid_keys);
  }
  if (plan_ == kUnplanned) AMD_RETURN_NOT_OK(Plan(t));
  if (plan_ == kDense) {
    bool left = false;
    AMD_ASSIGN_OR_RAISE(auto r, DenseStep(t, &left));
    if (!left) return r;
    // every rank read the same out-of-range count (it travelled inside the SUM): all of them redo this step on the
    // gather protocol and widen the plan at the next one
    ++replans_;
    int64_t lo, hi;
    AMD_RETURN_NOT_OK(KeyRange(t, &lo, &hi));
    if (hi >= lo) { seen_lo_ = have_seen_ ? std::min(lo, seen_lo_) : lo; seen_hi_ = have_seen_ ? std::max(hi, seen_hi_) : hi; have_seen_ = true; }
    plan_ = kUnplanned;
  }
  return GatherStep(t);
}

}  // namespace internal
}  // namespace compute
}  // namespace amdarrow
