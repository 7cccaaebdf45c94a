// sharded_groupby.h -- BASELINE config 5 in C++: Filter -> GroupBy{hash_sum, hash_count} over row-range shards, one
// rank (process or thread) per GPU, merged over RCCL / xGMI (the Python flavour is arrow-1_amd/dist.py).
//
// The reference has no counterpart (its GroupBy is single-process, hash_aggregate.cc:890-966); the contract is that the
// merged table every rank receives IS what internal::GroupBy returns over the concatenated shards, first-appearance
// order (:391-401) included.  No collective touches the data path: every rank runs the fused device group-by on its own
// rows; only the per-rank GROUP TABLES meet --
//   dense protocol  (integer keys of a small global range; the plan is made from one 16-byte all-reduce): two
//                   all-reduces per step, SUM over [sums | counts] and MIN over (owner rank << 32 | local id), ~24 MB
//                   for 1 M groups whatever the world size; the ordered read-out needs no sort;
//   gather protocol (anything else, and any step whose keys leave the dense plan -- that step is redone, the plan is
//                   widened): all-gather of the padded tables + re-aggregation with the same fused kernel.
// int64 sums are exact at any world size; float64 sums carry the usual reordering tolerance.
#pragma once

#include <memory>
#include <vector>

#include "amdarrow_compute.h"

namespace amdarrow {
namespace compute {
namespace internal {

// What the merge needs from the transport.  RCCL in production (MakeRcclCollectives, sharded_rccl.cc -- the only file
// of the host layer that includes RCCL / HIP headers); host_test substitutes threads-as-ranks on one GPU, because
// RCCL refuses two ranks per device.  Buffers are device memory; operations are in place and ordered on `stream`.
class Collectives {
 public:
  enum Op { SUM, MIN };
  enum Word { INT64, FLOAT64 };
  virtual ~Collectives() = default;
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual Status AllReduce(void* device_buf, int64_t count, Word word, Op op, void* stream) = 0;
  virtual Status AllGather(const void* device_send, void* device_recv, int64_t bytes_per_rank, void* stream) = 0;
};
// ncclComm_t of an initialised communicator (ncclCommInitRank), one rank per device
std::unique_ptr<Collectives> MakeRcclCollectives(void* nccl_comm, int rank, int world);
// a communicator of ONE rank on `device` (ncclCommInitAll): what a single-GPU run uses, and what host_test drives the
// RCCL calls themselves with (two ranks on one device are refused by RCCL)
Result<std::shared_ptr<Collectives>> MakeRcclSelfCollectives(int device);

class ShardedGroupBy {
 public:
  static Result<std::unique_ptr<ShardedGroupBy>> Make(ExecContext* ctx, std::shared_ptr<Collectives> coll, std::shared_ptr<DataType> key_type,
                                                      std::shared_ptr<DataType> value_type, int64_t group_capacity);
  // One step over this rank's row range (arrays in HBM; `mask` may be a null Datum).  Collective: every rank calls it
  // for the same step.  -> columns {hash_sum, hash_count, key_0}, identical on every rank.
  Result<GroupByResult> Run(const Datum& values, const Datum& keys, const Datum& mask);
  int replans() const { return replans_; }
  void set_merge_even_alone(bool on) { merge_alone_ = on; }  // world == 1 normally returns the local table; tests run the protocol anyway
  bool dense() const { return plan_ == kDense; }

 private:
  ShardedGroupBy() = default;
  struct Local { std::shared_ptr<Buffer> sums, counts, keys, key_validity; int64_t num_groups = 0; };
  Result<Local> LocalGroupBy(const ArrayData* values, const ArrayData& keys, const ArrayData* mask);
  Status KeyRange(const Local& t, int64_t* lo, int64_t* hi);  // collective
  Status Plan(const Local& t);                                  // collective
  Result<GroupByResult> DenseStep(const Local& t, bool* left_plan);
  Result<GroupByResult> GatherStep(const Local& t);
  GroupByResult Wrap(std::shared_ptr<Buffer> sums, std::shared_ptr<Buffer> counts, std::shared_ptr<Buffer> keys,
                     std::shared_ptr<Buffer> key_validity, int64_t num_groups, int64_t key_nulls);

  enum PlanKind { kUnplanned, kDense, kGather };
  ExecContext* ctx_ = nullptr;
  std::shared_ptr<Collectives> coll_;
  std::shared_ptr<DataType> key_type_, value_type_;
  int64_t capacity_ = 0;
  bool is_float_ = false;
  PlanKind plan_ = kUnplanned;
  int64_t kmin_ = 0, range_ = 0;
  bool have_seen_ = false;
  int64_t seen_lo_ = 0, seen_hi_ = -1;
  int replans_ = 0;
  bool merge_alone_ = false;
  std::shared_ptr<Buffer> acc_, owner_;
};

}  // namespace internal
}  // namespace compute
}  // namespace amdarrow
