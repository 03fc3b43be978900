// comm_impl.hip.h -- the exchange step of the multi-GPU path, included by ycnr_als.hip (after its
// fail() / HIP_TRY helpers, inside its anonymous namespace).
//
// The reference keeps every node's copy of the factor matrices current by streaming the rows of
// each finished portion to all peers over TCP ('alsSaveCalcedFactors', lib/emf/EmfMaster.js:711-723,
// receivers EmfLord.js:727-732 / EmfChief.js:207-212), copies both matrices to a joining node
// (EmfChief.js:55-71) and sends RMSE partial sums to the Lord ('rmseSaveCalcs', EmfMaster.js:726-736).
// Here one process drives one GPU and the same three roles are:
//   exchange      every rank's freshly solved row range goes straight into every other rank's replica
//                 of the matrix, at the rows' own offset -- no staging buffer, no copy in or out;
//   broadcast     one rank's whole matrix to all (join / warm start);
//   all-reduce    sums of doubles (RMSE partials).
// Transports:
//   YCNR_COMM_RCCL  the product path: RCCL over xGMI.  The exchange is ONE group of point-to-point
//                   ncclSend / ncclRecv between all pairs of ranks (a "direct" all-gather): xGMI is a
//                   full mesh of point-to-point links, so every link carries exactly the shard its two
//                   ends owe each other, all seven at once, instead of a ring passing every shard
//                   through every link in turn.  Uneven shards need no padding.  RCCL is loaded with
//                   dlopen when a communicator is created, so single-GPU users never load it.
//   YCNR_COMM_SHM   functional stand-in for tests: ranks on ONE node stage their rows through a POSIX
//                   shared-memory segment (synchronous, host barriers).  It exists so that several
//                   ranks can share one GPU (RCCL refuses duplicate devices) -- what gloo is to nccl.
//   YCNR_COMM_IPC   device-to-device without RCCL: every rank maps its peers' replicas (hipIpcOpenMemHandle;
//                   the handles travel through the same shared-memory segment, which is its control plane)
//                   and PUSHES its solved rows into them with hipMemcpyAsync on the communicator's stream --
//                   copy engines over xGMI, no compute units taken from the solve -- piece by piece behind
//                   the kernels that produced them; one host barrier at the end of the half-step says
//                   "every push has landed everywhere".  Unlike RCCL it runs with several ranks on one
//                   device, so the pipelined path (events, piece overlap, exposed-exchange accounting) is
//                   covered by the one-GPU tests.
//   YCNR_COMM_STUB  rank r of a world of N with the exchange left out: one GPU solves the shard of every
//                   rank in turn (bench.py --emulate-world), which gives the compute time per rank.
#pragma once
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>

struct RcclApi {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional: ycnr_als_comm_info reports what RCCL itself counts
};

// process-wide, loaded once
int rccl_api(const RcclApi **out) {
  static RcclApi api;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (!api.dl) {
    // librccl.so.1 is the SONAME of both ROCm's and PyTorch's copy: inside a torch process this
    // resolves to the library torch has already loaded (one RCCL per process)
    void *dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) return fail(YCNR_ERR_UNSUPPORTED, "RCCL is not available: %s", dlerror());
    RcclApi a;
    a.dl = dl;
#define YCNR_SYM(field, name)                                                            \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(dl, name));                        \
  if (!a.field) {                                                                        \
    dlclose(dl);                                                                         \
    return fail(YCNR_ERR_UNSUPPORTED, "RCCL: symbol %s not found", name);                \
  }
    YCNR_SYM(GetUniqueId, "ncclGetUniqueId")
    YCNR_SYM(CommInitRank, "ncclCommInitRank")
    YCNR_SYM(CommDestroy, "ncclCommDestroy")
    YCNR_SYM(GroupStart, "ncclGroupStart")
    YCNR_SYM(GroupEnd, "ncclGroupEnd")
    YCNR_SYM(Send, "ncclSend")
    YCNR_SYM(Recv, "ncclRecv")
    YCNR_SYM(AllReduce, "ncclAllReduce")
    YCNR_SYM(Broadcast, "ncclBroadcast")
    YCNR_SYM(GetErrorString, "ncclGetErrorString")
#undef YCNR_SYM
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(dl, "ncclCommCount"));
    api = a;
  }
  *out = &api;
  return YCNR_OK;
}

#define NCCL_TRY(api, expr)                                                                              \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return fail(YCNR_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString(r_)); \
  } while (0)

// header of the shared-memory segment of the SHM transport (zero-filled pages are its valid start state)
struct ShmHeader {
  std::atomic<uint32_t> count, gen;
  std::atomic<uint32_t> attached;
  uint32_t pad[13];
};
static_assert(sizeof(ShmHeader) == 64, "ShmHeader layout");

struct Comm {
  int transport = YCNR_COMM_NONE, rank = 0, world = 1;
  hipStream_t stream = nullptr;  // exchanges run here, next to the compute stream
  // RCCL
  const RcclApi *api = nullptr;
  ncclComm_t nccl = nullptr;
  double *dScratch = nullptr;  // all-reduce staging
  size_t scratchCount = 0;
  // SHM
  int fd = -1;
  char name[80] = {0};
  ShmHeader *hdr = nullptr;
  char *data = nullptr;
  size_t dataBytes = 0, mapBytes = 0;
  // IPC: peers' replicas of both factor matrices, mapped into this process
  struct IpcPeer {
    void *base[2] = {nullptr, nullptr};  // what hipIpcOpenMemHandle returned (closed with the communicator)
    char *fac[2] = {nullptr, nullptr};   // the peer's matrix of each side
  };
  std::vector<IpcPeer> peers;
  // IPC: one more buffer per side of every peer, mapped on demand (ipc_publish_extra): where the owner of a row receives the row's
  // band slabs (the item half-step sharded by user bands)
  struct IpcExtra {
    void *base = nullptr;
    char *ptr = nullptr;
  };
  std::vector<IpcExtra> extra[2];
  const void *mapped[2] = {nullptr, nullptr};  // this rank's matrices as the peers know them (re-published when rebound)
  bool pendingFinish = false;                 // pushes enqueued since the last end-of-step barrier

  bool active() const { return transport != YCNR_COMM_NONE && world > 1; }
};

int shm_barrier(Comm &c) {
  const uint32_t gen = c.hdr->gen.load(std::memory_order_acquire);
  if (c.hdr->count.fetch_add(1, std::memory_order_acq_rel) == (uint32_t)c.world - 1) {
    c.hdr->count.store(0, std::memory_order_relaxed);
    c.hdr->gen.fetch_add(1, std::memory_order_acq_rel);
    return YCNR_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; c.hdr->gen.load(std::memory_order_acquire) == gen; ++spins) {
    if ((spins & 1023) == 1023) {
      sched_yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
        return fail(YCNR_ERR_STATE, "shared-memory barrier: a peer did not arrive within 120 s (rank %d of %d)", c.rank, c.world);
    }
  }
  return YCNR_OK;
}

void ipc_close_extra(Comm &c, int side) {
  for (Comm::IpcExtra &e : c.extra[side]) {
    if (e.base) (void)hipIpcCloseMemHandle(e.base);
    e.base = nullptr;
    e.ptr = nullptr;
  }
}

void ipc_close_peers(Comm &c) {
  for (Comm::IpcPeer &p : c.peers)
    for (int s = 0; s < 2; ++s) {
      if (p.base[s]) (void)hipIpcCloseMemHandle(p.base[s]);
      p.base[s] = nullptr;
      p.fac[s] = nullptr;
    }
  c.mapped[0] = c.mapped[1] = nullptr;
}

void comm_release(Comm &c) {
  if (c.stream) (void)hipStreamSynchronize(c.stream);
  ipc_close_extra(c, 0);
  ipc_close_extra(c, 1);
  c.extra[0].clear();
  c.extra[1].clear();
  ipc_close_peers(c);
  c.peers.clear();
  c.pendingFinish = false;
  if (c.nccl && c.api) (void)c.api->CommDestroy(c.nccl);
  c.nccl = nullptr;
  if (c.dScratch) (void)hipFree(c.dScratch);
  c.dScratch = nullptr;
  if (c.hdr) munmap(c.hdr, c.mapBytes);
  c.hdr = nullptr;
  if (c.fd >= 0) close(c.fd);
  c.fd = -1;
  if (c.stream) {
    (void)hipStreamSynchronize(c.stream);
    (void)hipStreamDestroy(c.stream);
  }
  c.stream = nullptr;
  c.transport = YCNR_COMM_NONE;
  c.world = 1;
  c.rank = 0;
}

// dataBytes: the largest payload one exchange / broadcast / all-reduce of this job can stage (SHM only)
int comm_setup(Comm &c, const void *id, int transport, int rank, int world, size_t dataBytes) {
  if (world < 1 || rank < 0 || rank >= world) return fail(YCNR_ERR_INVALID, "comm_init: rank %d of %d", rank, world);
  if (!id) return fail(YCNR_ERR_INVALID, "comm_init: null id");
  c.rank = rank;
  c.world = world;
  HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  if (transport == YCNR_COMM_RCCL) {
    int rc = rccl_api(&c.api);
    if (rc) return rc;
    ncclUniqueId uid;
    static_assert(sizeof(uid) == YCNR_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(&uid, id, sizeof uid);
    NCCL_TRY(c.api, c.api->CommInitRank(&c.nccl, world, uid, rank));
    c.transport = YCNR_COMM_RCCL;
    return YCNR_OK;
  }
  if (transport == YCNR_COMM_STUB) {
    c.transport = YCNR_COMM_STUB;
    return YCNR_OK;
  }
  if (transport == YCNR_COMM_SHM || transport == YCNR_COMM_IPC) {
    // (IPC: the segment is the control plane -- handles and barriers -- and carries the all-reduce / broadcast)
    if (transport == YCNR_COMM_IPC) dataBytes = std::max<size_t>((size_t)16 << 20, (size_t)world * 256);
    const unsigned char *b = (const unsigned char *)id;
    snprintf(c.name, sizeof c.name, "/ycnr_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4], b[5], b[6],
             b[7], b[8], b[9], b[10], b[11]);
    c.dataBytes = dataBytes;
    c.mapBytes = sizeof(ShmHeader) + dataBytes;
    c.fd = shm_open(c.name, O_CREAT | O_RDWR, 0600);
    if (c.fd < 0) return fail(YCNR_ERR_HIP, "shm_open(%s) failed: %s", c.name, strerror(errno));
    if (ftruncate(c.fd, (off_t)c.mapBytes) != 0) return fail(YCNR_ERR_NOMEM, "ftruncate(%s, %zu) failed: %s", c.name, c.mapBytes, strerror(errno));
    void *p = mmap(nullptr, c.mapBytes, PROT_READ | PROT_WRITE, MAP_SHARED, c.fd, 0);
    if (p == MAP_FAILED) return fail(YCNR_ERR_NOMEM, "mmap(%s, %zu) failed: %s", c.name, c.mapBytes, strerror(errno));
    c.hdr = (ShmHeader *)p;
    c.data = (char *)p + sizeof(ShmHeader);
    c.transport = transport;
    if (transport == YCNR_COMM_IPC) c.peers.assign((size_t)world, Comm::IpcPeer());
    // the name can go once everybody has mapped the segment
    c.hdr->attached.fetch_add(1, std::memory_order_acq_rel);
    int rc = shm_barrier(c);
    if (rc) return rc;
    if (rank == 0) shm_unlink(c.name);
    return YCNR_OK;
  }
  return fail(YCNR_ERR_INVALID, "comm_init: unknown transport %d", transport);
}

// IPC: every rank publishes the handle of the allocation its two matrices live in (and their offsets inside it:
// a matrix may be a slice of a caching allocator's block) and maps its peers'.  Collective; called when a
// matrix has been (re)bound since the last publication.
struct IpcSlot {
  hipIpcMemHandle_t handle[2];
  uint64_t offset[2];
  int32_t device;
  int32_t status;  // 1: handles valid; -1: this rank failed before or while making them (every rank then returns an error)
};
// localFailed: the caller's own preparation of this collective call failed (e.g. the upload in front of the
// publication): the rank still takes part in both barriers and reports through its slot, so that its peers return an
// error at once instead of waiting 120 s at a barrier it never reaches, and the barrier counts stay in step.
int ipc_publish(Comm &c, void *const fac[2], bool localFailed = false) {
  if ((size_t)c.world * sizeof(IpcSlot) > c.dataBytes) return fail(YCNR_ERR_STATE, "ipc: control segment too small for %d ranks", c.world);
  IpcSlot mine;
  memset(&mine, 0, sizeof mine);
  mine.status = localFailed ? -1 : 1;
  std::string why;
  auto local_fail = [&](const char *what, hipError_t e) {
    if (mine.status > 0) why = std::string(what) + ": " + hipGetErrorString(e);
    mine.status = -1;
    (void)hipGetLastError();
  };
  if (mine.status > 0 && c.stream) {
    hipError_t e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) local_fail("hipStreamSynchronize", e);
  }
  ipc_close_peers(c);
  if (mine.status > 0) {
    hipError_t e = hipGetDevice(&mine.device);
    if (e != hipSuccess) local_fail("hipGetDevice", e);
  }
  for (int s = 0; s < 2 && mine.status > 0; ++s) {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)fac[s]);
    if (e != hipSuccess) {
      local_fail("hipMemGetAddressRange (is the matrix device memory of an ordinary allocation?)", e);
      break;
    }
    e = hipIpcGetMemHandle(&mine.handle[s], (void *)base);
    if (e != hipSuccess) {
      local_fail("hipIpcGetMemHandle", e);
      break;
    }
    mine.offset[s] = (uint64_t)((const char *)fac[s] - (const char *)base);
  }
  IpcSlot *slots = (IpcSlot *)c.data;
  memcpy(&slots[c.rank], &mine, sizeof mine);
  int rc = shm_barrier(c);
  if (rc) return rc;
  int failedRank = -1;
  for (int p = 0; p < c.world; ++p) {
    IpcSlot theirs;
    memcpy(&theirs, &slots[p], sizeof theirs);
    if (theirs.status <= 0 && failedRank < 0) failedRank = p;
  }
  std::string openErr;
  for (int p = 0; p < c.world && failedRank < 0 && openErr.empty(); ++p) {
    if (p == c.rank) continue;
    IpcSlot theirs;
    memcpy(&theirs, &slots[p], sizeof theirs);
    for (int s = 0; s < 2; ++s) {
      // both matrices of a peer may sit in ONE allocation (two slices of one block): map it once
      if (s == 1 && memcmp(&theirs.handle[0], &theirs.handle[1], sizeof(hipIpcMemHandle_t)) == 0) {
        c.peers[(size_t)p].fac[1] = (char *)c.peers[(size_t)p].base[0] + theirs.offset[1];
        continue;
      }
      void *base = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&base, theirs.handle[s], hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        char buf[200];
        snprintf(buf, sizeof buf, "hipIpcOpenMemHandle of rank %d's matrix %d failed: %s", p, s, hipGetErrorString(e));
        openErr = buf;
        break;
      }
      c.peers[(size_t)p].base[s] = base;
      c.peers[(size_t)p].fac[s] = (char *)base + theirs.offset[s];
    }
  }
  rc = shm_barrier(c);  // the slots may be overwritten again; everybody leaves through this barrier, failed or not
  if (rc) return rc;
  if (failedRank >= 0 || !openErr.empty()) {
    ipc_close_peers(c);
    if (failedRank == c.rank && !why.empty()) return fail(YCNR_ERR_HIP, "ipc: %s", why.c_str());
    if (failedRank >= 0) return fail(YCNR_ERR_STATE, "ipc: rank %d could not publish its matrices (this call failed on that rank)", failedRank);
    return fail(YCNR_ERR_HIP, "ipc: %s", openErr.c_str());
  }
  c.mapped[0] = fac[0];
  c.mapped[1] = fac[1];
  return YCNR_OK;
}

// IPC: every rank publishes one more device buffer of `side` (null: none) and maps its peers'.  Collective, same protocol as
// ipc_publish (a rank that failed locally still takes part and says so in its slot).
int ipc_publish_extra(Comm &c, int side, void *buf, bool localFailed = false) {
  if (c.transport != YCNR_COMM_IPC || c.world < 2) return YCNR_OK;
  if ((size_t)c.world * sizeof(IpcSlot) > c.dataBytes) return fail(YCNR_ERR_STATE, "ipc: control segment too small for %d ranks", c.world);
  ipc_close_extra(c, side);
  c.extra[side].assign((size_t)c.world, Comm::IpcExtra());
  IpcSlot mine;
  memset(&mine, 0, sizeof mine);
  mine.status = localFailed ? -1 : 1;
  std::string why;
  if (mine.status > 0 && buf) {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)buf);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&mine.handle[0], (void *)base);
    if (e != hipSuccess) {
      why = std::string("publishing the band-slab buffer: ") + hipGetErrorString(e);
      mine.status = -1;
      (void)hipGetLastError();
    } else {
      mine.offset[0] = (uint64_t)((const char *)buf - (const char *)base);
      mine.offset[1] = 1;  // a buffer is there
    }
  }
  IpcSlot *slots = (IpcSlot *)c.data;
  memcpy(&slots[c.rank], &mine, sizeof mine);
  int rc = shm_barrier(c);
  if (rc) return rc;
  int failedRank = -1;
  std::string openErr;
  for (int p = 0; p < c.world; ++p) {
    IpcSlot theirs;
    memcpy(&theirs, &slots[p], sizeof theirs);
    if (theirs.status <= 0 && failedRank < 0) failedRank = p;
  }
  for (int p = 0; p < c.world && failedRank < 0 && openErr.empty(); ++p) {
    if (p == c.rank) continue;
    IpcSlot theirs;
    memcpy(&theirs, &slots[p], sizeof theirs);
    if (!theirs.offset[1]) continue;
    void *base = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&base, theirs.handle[0], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      openErr = std::string("hipIpcOpenMemHandle of a peer's band-slab buffer failed: ") + hipGetErrorString(e);
      break;
    }
    c.extra[side][(size_t)p].base = base;
    c.extra[side][(size_t)p].ptr = (char *)base + theirs.offset[0];
  }
  rc = shm_barrier(c);
  if (rc) return rc;
  if (failedRank >= 0 || !openErr.empty()) {
    ipc_close_extra(c, side);
    if (failedRank == c.rank && !why.empty()) return fail(YCNR_ERR_HIP, "ipc: %s", why.c_str());
    if (failedRank >= 0) return fail(YCNR_ERR_STATE, "ipc: rank %d could not publish its band-slab buffer (this call failed on that rank)", failedRank);
    return fail(YCNR_ERR_HIP, "ipc: %s", openErr.c_str());
  }
  return YCNR_OK;
}

// IPC, before the first push of a collective call: every rank has entered it.  A push lands in the PEER's replica at
// the pusher's pace; whatever a peer's host did to its replica before it entered the call (ycnr_als_set_factors,
// a torch copy into the bound tensor) must be ordered before that -- with receives posted by the receiver (RCCL, SHM)
// this is implied, with pushes it is this barrier.
int ipc_enter(Comm &c) {
  if (c.transport != YCNR_COMM_IPC || c.world < 2) return YCNR_OK;
  return shm_barrier(c);
}

// IPC, end of a half-step: this rank's pushes have drained, then everybody's have (host barrier)
int ipc_finish(Comm &c) {
  if (c.transport != YCNR_COMM_IPC || !c.pendingFinish) return YCNR_OK;
  c.pendingFinish = false;
  HIP_TRY(hipStreamSynchronize(c.stream));
  return shm_barrier(c);
}

// Rows [begin[r], end[r]) of `fac` ([rows x k] elements of ts bytes) are current on rank r; bring
// every replica up to date.  `ready` has been recorded on the compute stream after the kernels
// that produced this rank's rows.  RCCL: enqueued on c.stream (returns at once; `done` is recorded
// behind it); SHM: synchronous.
// IPC: enqueued on c.stream like RCCL; ipc_finish (ycnr_als_sync) completes the half-step.  `side` selects the
// peers' mapping of this matrix.
int comm_exchange(Comm &c, void *fac, int side, int64_t k, size_t ts, const int64_t *begin, const int64_t *end, hipStream_t compute,
                  hipEvent_t ready, hipEvent_t t0, hipEvent_t t1, int64_t *bytesMoved) {
  const size_t rowBytes = (size_t)k * ts;
  int64_t moved = 0;
  for (int p = 0; p < c.world; ++p) {
    if (end[p] < begin[p]) return fail(YCNR_ERR_INVALID, "exchange: bad row range of rank %d", p);
    if (p != c.rank) moved += (end[p] - begin[p]) * (int64_t)rowBytes;                      // received
    else moved += (end[p] - begin[p]) * (int64_t)rowBytes * (int64_t)(c.world - 1);         // sent
  }
  if (bytesMoved) *bytesMoved += moved;
  if (c.transport == YCNR_COMM_RCCL) {
    HIP_TRY(hipEventRecord(ready, compute));
    HIP_TRY(hipStreamWaitEvent(c.stream, ready, 0));
    HIP_TRY(hipEventRecord(t0, c.stream));
    const ncclDataType_t dt = ts == 8 ? ncclDouble : ncclFloat;
    const size_t mine = (size_t)(end[c.rank] - begin[c.rank]) * (size_t)k;
    NCCL_TRY(c.api, c.api->GroupStart());
    // peers in "distance" order, so that at any moment the pairs (r, r + d) talk: every link busy
    for (int d = 1; d < c.world; ++d) {
      const int to = (c.rank + d) % c.world, from = (c.rank - d + c.world) % c.world;
      if (mine > 0) NCCL_TRY(c.api, c.api->Send((const char *)fac + (size_t)begin[c.rank] * rowBytes, mine, dt, to, c.nccl, c.stream));
      const size_t theirs = (size_t)(end[from] - begin[from]) * (size_t)k;
      if (theirs > 0) NCCL_TRY(c.api, c.api->Recv((char *)fac + (size_t)begin[from] * rowBytes, theirs, dt, from, c.nccl, c.stream));
    }
    NCCL_TRY(c.api, c.api->GroupEnd());
    HIP_TRY(hipEventRecord(t1, c.stream));
    return YCNR_OK;
  }
  if (c.transport == YCNR_COMM_STUB) {  // timing only: where the exchange would start and end
    HIP_TRY(hipEventRecord(ready, compute));
    HIP_TRY(hipStreamWaitEvent(c.stream, ready, 0));
    HIP_TRY(hipEventRecord(t0, c.stream));
    HIP_TRY(hipEventRecord(t1, c.stream));
    return YCNR_OK;
  }
  if (c.transport == YCNR_COMM_IPC) {
    if (c.mapped[side] != fac) return fail(YCNR_ERR_STATE, "ipc exchange: the matrix of side %d was rebound after it was published", side);
    HIP_TRY(hipEventRecord(ready, compute));
    HIP_TRY(hipStreamWaitEvent(c.stream, ready, 0));
    HIP_TRY(hipEventRecord(t0, c.stream));
    const size_t off = (size_t)begin[c.rank] * rowBytes, mine = (size_t)(end[c.rank] - begin[c.rank]) * rowBytes;
    if (mine > 0)
      for (int d = 1; d < c.world; ++d) {  // "distance" order, as the RCCL group: every link busy at once
        const int to = (c.rank + d) % c.world;
        HIP_TRY(hipMemcpyAsync(c.peers[(size_t)to].fac[side] + off, (const char *)fac + off, mine, hipMemcpyDeviceToDevice, c.stream));
      }
    HIP_TRY(hipEventRecord(t1, c.stream));
    c.pendingFinish = true;
    return YCNR_OK;
  }
  if (c.transport == YCNR_COMM_SHM) {
    HIP_TRY(hipEventRecord(t0, compute));
    HIP_TRY(hipStreamSynchronize(compute));
    int64_t lo = begin[0], hi = end[0];
    for (int p = 1; p < c.world; ++p) {
      lo = std::min(lo, begin[p]);
      hi = std::max(hi, end[p]);
    }
    if ((size_t)(hi - lo) * rowBytes > c.dataBytes) return fail(YCNR_ERR_STATE, "exchange: %zu bytes exceed the shared segment", (size_t)(hi - lo) * rowBytes);
    const size_t mine = (size_t)(end[c.rank] - begin[c.rank]) * rowBytes;
    if (mine) HIP_TRY(hipMemcpy(c.data + (size_t)(begin[c.rank] - lo) * rowBytes, (const char *)fac + (size_t)begin[c.rank] * rowBytes, mine, hipMemcpyDeviceToHost));
    int rc = shm_barrier(c);
    if (rc) return rc;
    for (int p = 0; p < c.world; ++p) {
      const size_t n = (size_t)(end[p] - begin[p]) * rowBytes;
      if (p != c.rank && n) HIP_TRY(hipMemcpy((char *)fac + (size_t)begin[p] * rowBytes, c.data + (size_t)(begin[p] - lo) * rowBytes, n, hipMemcpyHostToDevice));
    }
    rc = shm_barrier(c);  // the segment may be overwritten again
    if (rc) return rc;
    HIP_TRY(hipEventRecord(t1, compute));
    return YCNR_OK;
  }
  return fail(YCNR_ERR_STATE, "exchange without a communicator");
}

// vals[i] <- sum over ranks, in rank order (SHM) / RCCL's order
int comm_allreduce_sum(Comm &c, double *vals, int64_t n) {
  if (n <= 0 || !c.active() || c.transport == YCNR_COMM_STUB) return YCNR_OK;
  if (c.transport == YCNR_COMM_RCCL) {
    if (c.scratchCount < (size_t)n) {
      if (c.dScratch) (void)hipFree(c.dScratch);
      c.dScratch = nullptr;
      c.scratchCount = 0;
      HIP_TRY(hipMalloc(&c.dScratch, sizeof(double) * (size_t)n));
      c.scratchCount = (size_t)n;
    }
    HIP_TRY(hipMemcpyAsync(c.dScratch, vals, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c.stream));
    NCCL_TRY(c.api, c.api->AllReduce(c.dScratch, c.dScratch, (size_t)n, ncclDouble, ncclSum, c.nccl, c.stream));
    HIP_TRY(hipMemcpyAsync(vals, c.dScratch, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    return YCNR_OK;
  }
  if ((size_t)n * sizeof(double) * (size_t)c.world > c.dataBytes) return fail(YCNR_ERR_STATE, "all-reduce: %lld doubles exceed the shared segment", (long long)n);
  double *all = (double *)c.data;
  memcpy(all + (size_t)c.rank * (size_t)n, vals, sizeof(double) * (size_t)n);
  int rc = shm_barrier(c);
  if (rc) return rc;
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int p = 0; p < c.world; ++p) s += all[(size_t)p * (size_t)n + (size_t)i];
    vals[i] = s;
  }
  return shm_barrier(c);
}

// the whole matrix of `root` to every rank (a joining node's copy, lib/emf/EmfChief.js:55-71)
int comm_broadcast(Comm &c, void *fac, int side, size_t bytes, int root, hipStream_t compute) {
  if (!c.active() || c.transport == YCNR_COMM_STUB) return YCNR_OK;
  if (root < 0 || root >= c.world) return fail(YCNR_ERR_INVALID, "broadcast: root %d of %d", root, c.world);
  HIP_TRY(hipStreamSynchronize(compute));
  if (c.transport == YCNR_COMM_RCCL) {
    NCCL_TRY(c.api, c.api->Broadcast(fac, fac, bytes, ncclChar, root, c.nccl, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    return YCNR_OK;
  }
  if (c.transport == YCNR_COMM_IPC) {  // the root pushes into every replica
    if (c.mapped[side] != fac) return fail(YCNR_ERR_STATE, "ipc broadcast: the matrix of side %d was rebound after it was published", side);
    if (int rcb = ipc_enter(c)) return rcb;
    if (c.rank == root)
      for (int p = 0; p < c.world; ++p)
        if (p != root) HIP_TRY(hipMemcpyAsync(c.peers[(size_t)p].fac[side], fac, bytes, hipMemcpyDeviceToDevice, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    return shm_barrier(c);
  }
  if (bytes > c.dataBytes) return fail(YCNR_ERR_STATE, "broadcast: %zu bytes exceed the shared segment", bytes);
  if (c.rank == root) HIP_TRY(hipMemcpy(c.data, fac, bytes, hipMemcpyDeviceToHost));
  int rc = shm_barrier(c);
  if (rc) return rc;
  if (c.rank != root) HIP_TRY(hipMemcpy(fac, c.data, bytes, hipMemcpyHostToDevice));
  return shm_barrier(c);
}
