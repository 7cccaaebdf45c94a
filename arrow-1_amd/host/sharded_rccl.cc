// sharded_rccl.cc -- Collectives over RCCL (one rank per GPU, xGMI): the only file of the host layer that includes
// RCCL / HIP headers.  ncclAllReduce / ncclAllGather on the context's stream; nothing else.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "sharded_groupby.h"

namespace amdarrow {
namespace compute {
namespace internal {
namespace {
class RcclCollectives : public Collectives {
 public:
  RcclCollectives(ncclComm_t comm, int rank, int world) : comm_(comm), rank_(rank), world_(world) {}
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  Status AllReduce(void* buf, int64_t count, Word word, Op op, void* stream) override {
    const ncclResult_t r = ncclAllReduce(buf, buf, static_cast<size_t>(count), word == INT64 ? ncclInt64 : ncclFloat64,
                                         op == SUM ? ncclSum : ncclMin, comm_, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? Status::OK() : Status(StatusCode::UnknownError, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  }
  Status AllGather(const void* send, void* recv, int64_t bytes_per_rank, void* stream) override {
    const ncclResult_t r = ncclAllGather(send, recv, static_cast<size_t>(bytes_per_rank), ncclInt8, comm_, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? Status::OK() : Status(StatusCode::UnknownError, std::string("ncclAllGather: ") + ncclGetErrorString(r));
  }

 private:
  ncclComm_t comm_;
  int rank_, world_;
};
}  // namespace
Result<std::shared_ptr<Collectives>> MakeRcclSelfCollectives(int device) {
  ncclComm_t comm;
  int devs[1] = {device};
  const ncclResult_t r = ncclCommInitAll(&comm, 1, devs);
  if (r != ncclSuccess) return Status(StatusCode::UnknownError, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
  return std::shared_ptr<Collectives>(new RcclCollectives(comm, 0, 1));
}
std::unique_ptr<Collectives> MakeRcclCollectives(void* nccl_comm, int rank, int world) {
  return std::unique_ptr<Collectives>(new RcclCollectives(static_cast<ncclComm_t>(nccl_comm), rank, world));
}
}  // namespace internal
}  // namespace compute
}  // namespace amdarrow
