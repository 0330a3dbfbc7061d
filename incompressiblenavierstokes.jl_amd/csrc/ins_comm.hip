// Communication behind the C ABI (`ins_comm_create`, halo exchange, interface gather): RCCL over xGMI.
//
// The reference has no multi-device code; the z-slab decomposition of its stage loop is this library's own (§8e).
// A host (Julia with one task per GPU, MPI.jl ranks, or the Python harness) creates one communicator per GPU it drives:
//   * one process per GPU:   rank 0 calls ins_comm_unique_id, ships the 128 bytes to the other ranks by its own means, every rank calls
//                            ins_comm_create(nranks, rank, id) with its device current;
//   * one process, n GPUs:   ins_comm_create_local(n, devices, comms[]) (ncclCommInitAll) — what a single Julia process would use.
// Every exchange is a grouped ncclSend / ncclRecv set enqueued on the caller's stream: stream-ordered with the kernels that produce and
// consume the planes, no host synchronisation, nothing allocated per call.  librccl is loaded at run time (dlopen) the first time a
// communicator is created, so the single-GPU library has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "ins_internal.h"

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.so) return INS_OK;
  void* so = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (so) break;
  }
  if (!so) {
    ins_set_error("ins_comm: cannot load librccl (%s)", dlerror());
    return INS_ERR_UNSUPPORTED;
  }
#define INS_SYM(field, sym)                                            \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(so, #sym)); \
  if (!g_rccl.field) {                                                 \
    ins_set_error("ins_comm: librccl lacks %s", #sym);                 \
    return INS_ERR_UNSUPPORTED;                                        \
  }
  INS_SYM(GetUniqueId, ncclGetUniqueId)
  INS_SYM(CommInitRank, ncclCommInitRank)
  INS_SYM(CommInitAll, ncclCommInitAll)
  INS_SYM(CommDestroy, ncclCommDestroy)
  INS_SYM(GetErrorString, ncclGetErrorString)
  INS_SYM(GroupStart, ncclGroupStart)
  INS_SYM(GroupEnd, ncclGroupEnd)
  INS_SYM(Send, ncclSend)
  INS_SYM(Recv, ncclRecv)
  INS_SYM(AllReduce, ncclAllReduce)
  INS_SYM(AllGather, ncclAllGather)
#undef INS_SYM
  g_rccl.so = so;
  return INS_OK;
}

#define INS_NCCL_TRY(expr)                                                                              \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      ins_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));           \
      return INS_ERR_COMM;                                                                              \
    }                                                                                                   \
  } while (0)

// inside an open group: close it before returning, or the thread's group stays open and later calls are silently queued
#define INS_NCCL_TRY_G(expr)                                                                            \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      ins_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));           \
      g_rccl.GroupEnd();                                                                                \
      return INS_ERR_COMM;                                                                              \
    }                                                                                                   \
  } while (0)

}  // namespace

struct ins_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
};

extern "C" int ins_comm_unique_id(void* id128) {
  INS_REQUIRE(id128, "null argument");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  INS_NCCL_TRY(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == INS_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return INS_OK;
}

extern "C" int ins_comm_create(int nranks, int rank, const void* id128, ins_comm_t** out) {
  INS_REQUIRE(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ins_comm* c = new ins_comm();
  c->nranks = nranks;
  c->rank = rank;
  INS_HIP_TRY(hipGetDevice(&c->device));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    ins_set_error("ncclCommInitRank(rank %d of %d): %s", rank, nranks, g_rccl.GetErrorString(r));
    delete c;
    return INS_ERR_COMM;
  }
  *out = c;
  return INS_OK;
}

// One host thread driving several communicators (ins_comm_create_local with ngpu > 1): RCCL wants the calls of ALL its devices inside one group, so the host
// brackets each round of per-communicator exchange calls with these two (the groups the entry points open themselves nest inside).
extern "C" int ins_comm_group_begin(void) {
  int rc = load_rccl();
  if (rc) return rc;
  INS_NCCL_TRY(g_rccl.GroupStart());
  return INS_OK;
}
extern "C" int ins_comm_group_end(void) {
  int rc = load_rccl();
  if (rc) return rc;
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}

extern "C" int ins_comm_create_local(int ngpu, const int* devices, ins_comm_t** out) {
  INS_REQUIRE(out && ngpu >= 1 && ngpu <= 64, "bad argument");
  int rc = load_rccl();
  if (rc) return rc;
  std::vector<ncclComm_t> comms(ngpu);
  std::vector<int> devs(ngpu);
  for (int i = 0; i < ngpu; ++i) devs[i] = devices ? devices[i] : i;
  INS_NCCL_TRY(g_rccl.CommInitAll(comms.data(), ngpu, devs.data()));
  for (int i = 0; i < ngpu; ++i) {
    ins_comm* c = new ins_comm();
    c->comm = comms[i];
    c->nranks = ngpu;
    c->rank = i;
    c->device = devs[i];
    out[i] = c;
  }
  return INS_OK;
}

extern "C" int ins_comm_destroy(ins_comm_t* c) {
  if (!c) return INS_OK;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return INS_OK;
}

extern "C" int ins_comm_rank(const ins_comm_t* c, int* rank, int* nranks) {
  INS_REQUIRE(c && rank && nranks, "null argument");
  *rank = c->rank;
  *nranks = c->nranks;
  return INS_OK;
}

// One grouped exchange: nsend sends and nrecv receives of fp64 elements.  Messages between one pair of ranks match in posting order
// (the P = 2 case, where the lower and the upper neighbour are the same peer, relies on that: tests/test_gpu_comm_abi.py).
extern "C" int ins_comm_sendrecv_f64(ins_comm_t* c, int nsend, const double* const* sendbufs, const int64_t* sendcounts, const int32_t* dsts,
                                     int nrecv, double* const* recvbufs, const int64_t* recvcounts, const int32_t* srcs, void* stream) {
  INS_REQUIRE(c && nsend >= 0 && nrecv >= 0, "bad argument");
  INS_REQUIRE((nsend == 0 || (sendbufs && sendcounts && dsts)) && (nrecv == 0 || (recvbufs && recvcounts && srcs)), "null argument");
  for (int i = 0; i < nsend; ++i) INS_REQUIRE(sendbufs[i] && sendcounts[i] >= 0 && dsts[i] >= 0 && dsts[i] < c->nranks, "bad send");
  for (int i = 0; i < nrecv; ++i) INS_REQUIRE(recvbufs[i] && recvcounts[i] >= 0 && srcs[i] >= 0 && srcs[i] < c->nranks, "bad receive");
  hipStream_t s = as_stream(stream);
  INS_NCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < nsend; ++i) INS_NCCL_TRY_G(g_rccl.Send(sendbufs[i], (size_t)sendcounts[i], ncclDouble, dsts[i], c->comm, s));
  for (int i = 0; i < nrecv; ++i) INS_NCCL_TRY_G(g_rccl.Recv(recvbufs[i], (size_t)recvcounts[i], ncclDouble, srcs[i], c->comm, s));
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}

// z ghost planes of a padded local vector field `u` on a slab grid (z sides INS_BC_HALO): plane nzl -> next rank's plane 0 and, unless
// down_only, plane 1 -> previous rank's plane nzl + 1; components selected by comp_mask (bit c).  Planes span the whole padded (x, y)
// extent, so edges and corners stay consistent (boundary_conditions.jl:97-103).  One group call.
extern "C" int ins_halo_exchange_f64(ins_comm_t* c, const ins_grid_t* G, double* u, int comp_mask, int down_only, void* stream) {
  INS_REQUIRE(c && G && u, "null argument");
  const GridDev& g = G->g;
  INS_REQUIRE(g.D == 3 && g.bc[2][0] == INS_BC_HALO && g.bc[2][1] == INS_BC_HALO, "halo exchange needs a slab grid (INS_BC_HALO in z)");
  const long long plane = (long long)g.N[0] * g.N[1];
  const int nzl = g.N[2] - 2;
  const int prev = (c->rank + c->nranks - 1) % c->nranks, next = (c->rank + 1) % c->nranks;
  hipStream_t s = as_stream(stream);
  INS_NCCL_TRY(g_rccl.GroupStart());
  // posting order: everything for `next`, then everything for `prev`; receives: from `prev`, then from `next` (pairs up when prev == next)
  for (int a = 0; a < 3; ++a)
    if (comp_mask & (1 << a)) INS_NCCL_TRY_G(g_rccl.Send(u + a * g.sc + (long long)nzl * plane, (size_t)plane, ncclDouble, next, c->comm, s));
  if (!down_only)
    for (int a = 0; a < 3; ++a)
      if (comp_mask & (1 << a)) INS_NCCL_TRY_G(g_rccl.Send(u + a * g.sc + plane, (size_t)plane, ncclDouble, prev, c->comm, s));
  for (int a = 0; a < 3; ++a)
    if (comp_mask & (1 << a)) INS_NCCL_TRY_G(g_rccl.Recv(u + a * g.sc, (size_t)plane, ncclDouble, prev, c->comm, s));
  if (!down_only)
    for (int a = 0; a < 3; ++a)
      if (comp_mask & (1 << a)) INS_NCCL_TRY_G(g_rccl.Recv(u + a * g.sc + (long long)(nzl + 1) * plane, (size_t)plane, ncclDouble, next, c->comm, s));
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}

// Ghost planes of the extended pressure buffer [1 plane below | nzl local planes | 2 planes above] (unpadded n0 x n1 planes): my last local
// plane -> next rank's `below`; my first two local planes -> previous rank's two `above` planes.
extern "C" int ins_halo_exchange_p_f64(ins_comm_t* c, double* p_ext, int64_t plane_elems, int nzl, void* stream) {
  INS_REQUIRE(c && p_ext && plane_elems > 0 && nzl >= 2, "bad argument");
  const int prev = (c->rank + c->nranks - 1) % c->nranks, next = (c->rank + 1) % c->nranks;
  hipStream_t s = as_stream(stream);
  const size_t pl = (size_t)plane_elems;
  INS_NCCL_TRY(g_rccl.GroupStart());
  INS_NCCL_TRY_G(g_rccl.Send(p_ext + pl * nzl, pl, ncclDouble, next, c->comm, s));
  INS_NCCL_TRY_G(g_rccl.Send(p_ext + pl, 2 * pl, ncclDouble, prev, c->comm, s));
  INS_NCCL_TRY_G(g_rccl.Recv(p_ext, pl, ncclDouble, prev, c->comm, s));
  INS_NCCL_TRY_G(g_rccl.Recv(p_ext + pl * (nzl + 1), 2 * pl, ncclDouble, next, c->comm, s));
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}

// Interface data of the distributed tridiagonal z solve (csrc/ins_ztri.hip): edges_all[r * count .. (r + 1) * count) = rank r's `edge`.
// direct != 0: one send to every peer in one group — xGMI is a full mesh, each block crosses exactly one link and all links work at
// once; 0: ncclAllGather (ring).
extern "C" int ins_ztri_allgather_f64(ins_comm_t* c, const double* edge, double* edges_all, int64_t count, int direct, void* stream) {
  INS_REQUIRE(c && edge && edges_all && count >= 0, "bad argument");
  hipStream_t s = as_stream(stream);
  if (!direct) {
    INS_NCCL_TRY(g_rccl.AllGather(edge, edges_all, (size_t)count, ncclDouble, c->comm, s));
    return INS_OK;
  }
  INS_HIP_TRY(hipMemcpyAsync(edges_all + (long long)c->rank * count, edge, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (c->nranks == 1) return INS_OK;
  INS_NCCL_TRY(g_rccl.GroupStart());
  for (int d = 1; d < c->nranks; ++d) INS_NCCL_TRY_G(g_rccl.Send(edge, (size_t)count, ncclDouble, (c->rank + d) % c->nranks, c->comm, s));
  for (int d = c->nranks - 1; d >= 1; --d) {
    const int q = (c->rank + d) % c->nranks;
    INS_NCCL_TRY_G(g_rccl.Recv(edges_all + (long long)q * count, (size_t)count, ncclDouble, q, c->comm, s));
  }
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}

// In-place scalar reductions across ranks (CFL minimum, norms, energies — what `ncclAllReduce` is for in SURVEY.md §8e): op 0 sum, 1 max, 2 min.
extern "C" int ins_comm_allreduce_f64(ins_comm_t* c, double* buf, int64_t count, int op, void* stream) {
  INS_REQUIRE(c && buf && count >= 0 && op >= 0 && op <= 2, "bad argument");
  const ncclRedOp_t o = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
  INS_NCCL_TRY(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, o, c->comm, as_stream(stream)));
  return INS_OK;
}

int ins_comm_allreduce_internal(ins_comm* c, double* buf, long long count, int op, hipStream_t s) {
  return ins_comm_allreduce_f64(c, buf, count, op, s);
}

// z ghost planes of a padded local SCALAR field (pressure-like: the search direction of the slab CG): plane nzl -> next rank's plane 0,
// plane 1 -> previous rank's plane nzl + 1 — the periodic z part of apply_bc_p! (boundary_conditions.jl:306-318) across ranks.
extern "C" int ins_halo_exchange_scalar_f64(ins_comm_t* c, const ins_grid_t* G, double* p, void* stream) {
  INS_REQUIRE(c && G && p, "null argument");
  const GridDev& g = G->g;
  INS_REQUIRE(g.D == 3 && g.bc[2][0] == INS_BC_HALO && g.bc[2][1] == INS_BC_HALO, "halo exchange needs a slab grid (INS_BC_HALO in z)");
  const size_t plane = (size_t)g.N[0] * g.N[1];
  const int nzl = g.N[2] - 2;
  const int prev = (c->rank + c->nranks - 1) % c->nranks, next = (c->rank + 1) % c->nranks;
  hipStream_t s = as_stream(stream);
  INS_NCCL_TRY(g_rccl.GroupStart());
  INS_NCCL_TRY_G(g_rccl.Send(p + plane * nzl, plane, ncclDouble, next, c->comm, s));
  INS_NCCL_TRY_G(g_rccl.Send(p + plane, plane, ncclDouble, prev, c->comm, s));
  INS_NCCL_TRY_G(g_rccl.Recv(p, plane, ncclDouble, prev, c->comm, s));
  INS_NCCL_TRY_G(g_rccl.Recv(p + plane * (nzl + 1), plane, ncclDouble, next, c->comm, s));
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}
int ins_comm_halo_scalar_internal(ins_comm* c, const ins_grid* G, double* p, hipStream_t s) { return ins_halo_exchange_scalar_f64(c, G, p, s); }

// Transpose exchange of the z-FFT route: block r of `send` (count elements) -> rank r, whose block `rank` of `recv` receives it.
extern "C" int ins_comm_alltoall_f64(ins_comm_t* c, const double* send, double* recv, int64_t count, void* stream) {
  INS_REQUIRE(c && send && recv && count >= 0, "bad argument");
  hipStream_t s = as_stream(stream);
  INS_NCCL_TRY(g_rccl.GroupStart());
  for (int r = 0; r < c->nranks; ++r) {
    INS_NCCL_TRY_G(g_rccl.Send(send + (long long)r * count, (size_t)count, ncclDouble, r, c->comm, s));
    INS_NCCL_TRY_G(g_rccl.Recv(recv + (long long)r * count, (size_t)count, ncclDouble, r, c->comm, s));
  }
  INS_NCCL_TRY(g_rccl.GroupEnd());
  return INS_OK;
}
