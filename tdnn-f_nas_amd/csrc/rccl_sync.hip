// rccl_sync.hip -- the data-parallel exchange steps on RCCL, called from C++ on the library's own streams.
//
// egs minibatches shard over the GPUs of a node; the exchanges are (1) one all-reduce (sum) of the flat raw
// parameter-gradient buffer per minibatch, bucketed and overlapped with the backward pass, and (2) for exact single-GPU equivalence the
// column sums of every train-mode BatchNorm (/root/reference/src/nnet3/nnet-normalize-component.cc:433-445 takes its statistics over all
// rows of the minibatch).  Both are enqueued here with ncclAllReduce -- the gradient buckets on a communication stream behind the event
// net.hip records when a bucket is final, the BatchNorm sums on the compute stream between the two finalize launches -- so that no
// host language sits in the step's critical path (round 3 called back into Python 56 times per step for the BatchNorm sums).
//
// librccl.so is resolved with dlopen on first use: the library has no link-time dependency on RCCL, a single-GPU user never loads it.
// The communicator is created here from a unique id the caller distributes (tdnnf_rccl_unique_id on rank 0 -> every rank's
// tdnnf_rccl_comm_create): one process per GPU, as torch.distributed's launcher starts them.
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "net.h"

namespace {

struct UniqueId {
  char internal[128];  // NCCL_UNIQUE_ID_BYTES
};
typedef void *Comm;
typedef int (*get_unique_id_t)(UniqueId *);
typedef int (*comm_init_rank_t)(Comm *, int, UniqueId, int);
typedef int (*comm_destroy_t)(Comm);
typedef int (*all_reduce_t)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef const char *(*get_error_string_t)(int);

struct Rccl {
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  all_reduce_t all_reduce = nullptr;
  get_error_string_t error_string = nullptr;
  bool ok = false;
  const char *how = "not found";
};
// Which RCCL: a process that already has one mapped (torch.distributed's "nccl" backend brings torch/lib/librccl.so) must use THAT
// copy -- a second librccl.so in the process means two sets of proxy threads, two IPC caches and two views of the topology.  So:
// (1) any loaded object whose path names librccl, re-opened with RTLD_NOLOAD; (2) the global symbol scope; (3) only when the process
// has none, dlopen by name.  tdnnf_rccl_library_path reports what was taken.
struct FindLoaded {
  char path[1024];
};
int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
    snprintf(((FindLoaded *)data)->path, sizeof(FindLoaded::path), "%s", info->dlpi_name);
    return 1;
  }
  return 0;
}
bool bind(Rccl &x, void *h) {
  x.get_unique_id = (get_unique_id_t)dlsym(h, "ncclGetUniqueId");
  x.comm_init_rank = (comm_init_rank_t)dlsym(h, "ncclCommInitRank");
  x.comm_destroy = (comm_destroy_t)dlsym(h, "ncclCommDestroy");
  x.all_reduce = (all_reduce_t)dlsym(h, "ncclAllReduce");
  x.error_string = (get_error_string_t)dlsym(h, "ncclGetErrorString");
  x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce;
  return x.ok;
}
const Rccl &rccl() {
  static const Rccl r = [] {
    Rccl x;
    FindLoaded f;
    f.path[0] = 0;
    if (dl_iterate_phdr(find_loaded_rccl, &f)) {
      void *h = dlopen(f.path, RTLD_NOW | RTLD_NOLOAD);
      if (h && bind(x, h)) {
        x.how = "already mapped in the process";
        return x;
      }
    }
    if (bind(x, RTLD_DEFAULT)) {
      x.how = "global symbol scope";
      return x;
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
      void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h && bind(x, h)) {
        x.how = "dlopen by name (the process had none)";
        break;
      }
    }
    return x;
  }();
  return r;
}
constexpr int kNcclSum = 0, kNcclFloat = 7, kNcclDouble = 8;

int nccl_status(int rc, const char *what) {
  if (rc == 0) return TDNNF_OK;
  tdnnf::set_error("RCCL error %d (%s) in %s", rc, rccl().error_string ? rccl().error_string(rc) : "?", what);
  return TDNNF_EHIP;
}

// the BatchNorm hook (common.h BnSync::fn): ctx is the communicator
int bn_allreduce(void *ctx, double *buf, long long count, tdnnf_stream stream) {
  return rccl().all_reduce(buf, buf, (size_t)count, kNcclDouble, kNcclSum, (Comm)ctx, (hipStream_t)stream) == 0 ? 0 : 1;
}

}  // namespace

using namespace tdnnf;

extern "C" {

int tdnnf_rccl_available(void) { return rccl().ok ? 1 : 0; }

int tdnnf_rccl_library_path(char *out, int out_bytes) {
  TDNNF_REQUIRE(out && out_bytes > 0, "rccl_library_path: bad arguments");
  out[0] = 0;
  if (!rccl().ok) return TDNNF_OK;
  Dl_info di;
  memset(&di, 0, sizeof(di));
  if (dladdr((void *)rccl().all_reduce, &di) && di.dli_fname) snprintf(out, (size_t)out_bytes, "%s [%s]", di.dli_fname, rccl().how);
  else snprintf(out, (size_t)out_bytes, "? [%s]", rccl().how);
  return TDNNF_OK;
}

int tdnnf_rccl_unique_id(void *out_128_bytes) {
  TDNNF_REQUIRE(out_128_bytes, "rccl_unique_id: null argument");
  TDNNF_REQUIRE(rccl().ok, "rccl_unique_id: librccl.so is not available");
  UniqueId id;
  memset(&id, 0, sizeof(id));
  int rc = nccl_status(rccl().get_unique_id(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(out_128_bytes, &id, sizeof(id));
  return TDNNF_OK;
}

int tdnnf_rccl_comm_create(const void *id_128_bytes, int world_size, int rank, void **comm_out) {
  TDNNF_REQUIRE(id_128_bytes && comm_out && world_size >= 1 && rank >= 0 && rank < world_size, "rccl_comm_create: bad arguments");
  TDNNF_REQUIRE(rccl().ok, "rccl_comm_create: librccl.so is not available");
  UniqueId id;
  memcpy(&id, id_128_bytes, sizeof(id));
  Comm c = nullptr;
  int rc = nccl_status(rccl().comm_init_rank(&c, world_size, id, rank), "ncclCommInitRank");
  if (rc) return rc;
  *comm_out = c;
  return TDNNF_OK;
}

void tdnnf_rccl_comm_destroy(void *comm) {
  if (comm && rccl().ok) rccl().comm_destroy((Comm)comm);
}

int tdnnf_rccl_allreduce_sum(void *comm, void *buf_dev, long long count, int is_double, tdnnf_stream stream) {
  TDNNF_REQUIRE(comm && (buf_dev || count == 0) && count >= 0, "rccl_allreduce_sum: bad arguments");
  TDNNF_REQUIRE(rccl().ok, "rccl_allreduce_sum: librccl.so is not available");
  if (count == 0) return TDNNF_OK;
  return nccl_status(rccl().all_reduce(buf_dev, buf_dev, (size_t)count, is_double ? kNcclDouble : kNcclFloat, kNcclSum, (Comm)comm, (hipStream_t)stream),
                     "ncclAllReduce");
}

int tdnnf_net_set_batchnorm_sync_rccl(tdnnf_net *n, void *comm, int world_size) {
  TDNNF_REQUIRE(n && world_size >= 1, "net_set_batchnorm_sync_rccl: bad arguments");
  if (!comm) return tdnnf_net_set_batchnorm_sync(n, nullptr, nullptr, 1);
  TDNNF_REQUIRE(rccl().ok, "net_set_batchnorm_sync_rccl: librccl.so is not available");
  return tdnnf_net_set_batchnorm_sync(n, bn_allreduce, comm, world_size);
}

// The gradient exchange of one minibatch: every bucket of the flat gradient buffer is summed over the ranks on `comm_stream`, each
// behind the event the backward pass recorded when the bucket became final (tdnnf_net_grad_bucket), so the upper layers' reductions
// run under the lower layers' backward pass; `stream` (the compute stream) then waits for the last one.  Call right after
// tdnnf_net_forward_backward.
int tdnnf_net_allreduce_grads_rccl(tdnnf_net *n, void *comm, tdnnf_stream comm_stream, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && n->grads && comm && comm_stream, "net_allreduce_grads_rccl: bad arguments (a communication stream of its own is required)");
  TDNNF_REQUIRE(rccl().ok, "net_allreduce_grads_rccl: librccl.so is not available");
  for (auto &gb : n->buckets) {
    TDNNF_HIP(hipStreamWaitEvent((hipStream_t)comm_stream, gb.ready, 0));
    const long long cnt = gb.end - gb.begin;
    if (cnt <= 0) continue;
    int rc = nccl_status(rccl().all_reduce(n->grads + gb.begin, n->grads + gb.begin, (size_t)cnt, kNcclFloat, kNcclSum, (Comm)comm, (hipStream_t)comm_stream),
                         "ncclAllReduce (gradient bucket)");
    if (rc) return rc;
  }
  if (!n->ev_comm) TDNNF_HIP(hipEventCreateWithFlags(&n->ev_comm, hipEventDisableTiming));
  TDNNF_HIP(hipEventRecord(n->ev_comm, (hipStream_t)comm_stream));
  TDNNF_HIP(hipStreamWaitEvent((hipStream_t)stream, n->ev_comm, 0));
  return TDNNF_OK;
}

}  // extern "C"
