// transport.hip -- the library's own message transport under the exchange points (include/pomgpu.h,
// "transport"): what the reference does with MPI_SEND / MPI_RECV inside exchange2d_mpi / exchange3d_mpi /
// order2d_mpi / order3d_mpi (parallel_mpi.f:154-480) and with MPI_INIT / MPI_COMM_RANK / MPI_COMM_SIZE in
// initialize_mpi (parallel_mpi.f:124-151).
//
// Production mover: RCCL.  One message round = ONE ncclGroupStart .. ncclGroupEnd with an ncclSend and an
// ncclRecv per neighbour (up to eight), enqueued on the stream the kernels run on, so that pack kernel ->
// messages -> unpack kernel need no event and no host synchronisation.  Between the GPUs of one node the
// messages travel over xGMI (point-to-point links: each neighbour has its own link, a round with four
// neighbours uses four links at once).  librccl is opened with dlopen at run time: a process that never asks
// for the RCCL transport (one tile; tests) does not load it, and a Python host passes the path of the
// librccl that torch already mapped so that one process never holds two copies.
//
// Test mover: a host callback (pomgpu_set_transport) -- ranks that share one GPU cannot talk RCCL to each
// other (it refuses two ranks on one device), host threads driving the CPU build of the kernels have no GPU.
#include <stdio.h>
#include <string.h>
#include "pomgpu.h"
#include "pomgpu_internal.hpp"

#ifndef POMGPU_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enumerators only; no symbol of librccl is linked

struct RcclApi {
  void *lib;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *);   // optional (RCCL >= 2.18): the side stream's communicator
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);   // the ranks' agreement on side-stream rounds
  ncclResult_t (*CommCount)(const ncclComm_t, int *);         // optional: how many ranks the communicator really has (pomgpu_rccl_nranks)
};
static RcclApi g_rccl;
struct RcclComm { ncclComm_t comm, comm2; int rank, nranks; };

static int rccl_load(const char *path) {
  if (g_rccl.lib) return 0;
  const char *p = (path && path[0]) ? path : "librccl.so";
  void *h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  if (!h && !(path && path[0])) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "pomgpu: cannot open %s: %s\n", p, dlerror()); return -1; }
  RcclApi a;
  memset(&a, 0, sizeof a);
  a.lib = h;
#define SYM(field, name)                                                                  \
  *(void **)(&a.field) = dlsym(h, name);                                                  \
  if (!a.field) { fprintf(stderr, "pomgpu: %s lacks %s\n", p, name); dlclose(h); return -1; }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(AllReduce, "ncclAllReduce")
#undef SYM
  *(void **)(&a.CommSplit) = dlsym(h, "ncclCommSplit");       // absent in old libraries: no side-stream rounds then
  *(void **)(&a.CommCount) = dlsym(h, "ncclCommCount");
  g_rccl = a;
  return 0;
}
#endif

void pomgpu_tp_free(pomgpu_ctx *c) {
  pomgpu_transport &T = c->tp;
#ifndef POMGPU_EMU
  if (T.rccl) {
    RcclComm *r = (RcclComm *)T.rccl;
    if (g_rccl.lib && r->comm2) (void)g_rccl.CommDestroy(r->comm2);
    if (g_rccl.lib && r->comm) (void)g_rccl.CommDestroy(r->comm);
    delete r;
    T.rccl = NULL;
  }
#endif
  for (int d = 0; d < 8; d++) {
    (void)hipFree(T.send[d]); (void)hipFree(T.recv[d]);
    T.send[d] = T.recv[d] = NULL; T.cap[d] = 0;
    (void)hipFree(T.send2[d]); (void)hipFree(T.recv2[d]);
    T.send2[d] = T.recv2[d] = NULL; T.cap2[d] = 0;
  }
  T.on = 0;
  T.side_agreed = 0;
  T.wr_side = 0;
  T.fn_ordered = 0;
}

// grow the staging buffers to at least need[d] doubles (directions without a neighbour stay empty)
int pomgpu_tp_reserve(pomgpu_ctx *c, const size_t *need) {
  pomgpu_transport &T = c->tp;
  for (int d = 0; d < 8; d++) {
    if (T.nbr[d] < 0 || need[d] <= T.cap[d]) continue;
    if (T.cap[d]) (void)hipStreamSynchronize(c->stream);      // the old buffers may still be in flight
    (void)hipFree(T.send[d]); (void)hipFree(T.recv[d]);
    T.send[d] = T.recv[d] = NULL; T.cap[d] = 0;
    if (hipMalloc((void **)&T.send[d], need[d] * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&T.recv[d], need[d] * sizeof(double)) != hipSuccess)
      return pomgpu_fail(c, POMGPU_ENOMEM, "transport: cannot allocate %zu-byte staging buffers", need[d] * sizeof(double));
    T.cap[d] = need[d];
  }
  return POMGPU_OK;
}

int pomgpu_tp_reserve2(pomgpu_ctx *c, const size_t *need) {
  pomgpu_transport &T = c->tp;
  for (int d = 0; d < 8; d++) {
    if (T.nbr[d] < 0 || need[d] <= T.cap2[d]) continue;
    if (T.cap2[d] && c->side) (void)hipStreamSynchronize(c->side);
    (void)hipFree(T.send2[d]); (void)hipFree(T.recv2[d]);
    T.send2[d] = T.recv2[d] = NULL; T.cap2[d] = 0;
    if (hipMalloc((void **)&T.send2[d], need[d] * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&T.recv2[d], need[d] * sizeof(double)) != hipSuccess)
      return pomgpu_fail(c, POMGPU_ENOMEM, "transport: cannot allocate %zu-byte staging buffers (side stream)", need[d] * sizeof(double));
    T.cap2[d] = need[d];
  }
  return POMGPU_OK;
}
// Rounds on the side stream need a communicator of their own (one communicator serves one stream at a time) and the
// consent of EVERY rank: T.side_agreed is the minimum over the ranks of "I could" (pomgpu_tp_rccl reduces it over the
// communicator; with a callback mover the host reduces pomgpu_transport_side_capable and reports through
// pomgpu_transport_side_agree).  A rank deciding for itself would post its rounds on another communicator and in another
// order than its neighbours: a deadlock.
int pomgpu_tp_side_ok(pomgpu_ctx *c) {
  pomgpu_transport &T = c->tp;
  if (!T.on || !c->side || !T.side_agreed) return 0;
  if (T.fn) return 1;
#ifndef POMGPU_EMU
  RcclComm *r = (RcclComm *)T.rccl;
  return r && r->comm2 != NULL;
#else
  return 0;
#endif
}
// this rank's own answer (not yet an agreement): a second stream exists or can be made, nothing switches the overlap off
static int side_capable_local(pomgpu_ctx *c) {
  if (SW(c, NO_OVERLAP) || SW(c, NO_SIDE_COMM)) return 0;      // this context's switches (read at pomgpu_create)
  return pomgpu_side_stream(c);
}
// The answer a host reduces over ALL ranks with min (callback movers: the library cannot reach the other ranks itself):
// 0 = this rank cannot serve side-stream rounds, 1 = it can but wants the wr exchange on the main stream (WR_MAIN),
// 2 = it can and wr may run beside the next step.  The minimum is what pomgpu_transport_side_agree takes.
extern "C" int pomgpu_transport_side_capable(pomgpu_ctx *c) {
  if (!c || !c->tp.on) return 0;
#ifndef POMGPU_EMU
  if (!c->tp.fn) { RcclComm *r = (RcclComm *)c->tp.rccl; if (!r || !r->comm2) return 0; }
#endif
  if (!side_capable_local(c)) return 0;
  return SW(c, WR_MAIN) ? 1 : 2;
}
// the digest of the switches every rank of a decomposition must share (pomgpu_internal.hpp): hosts with a callback mover
// compare it over their ranks before the first step (RCCL: pomgpu_rccl_init does it over the communicator)
extern "C" int pomgpu_transport_stream_ordered(pomgpu_ctx *c, int ordered) {
  if (!c) return POMGPU_EINVAL;
  if (!c->tp.on || !c->tp.fn) return pomgpu_fail(c, POMGPU_EINVAL, "transport_stream_ordered: set a callback transport first");
  c->tp.fn_ordered = ordered ? 1 : 0;
  return POMGPU_OK;
}
extern "C" unsigned pomgpu_switch_digest(pomgpu_ctx *c) { return c ? pomgpu_switches_collective_digest(c->sw) : 0u; }
extern "C" int pomgpu_transport_side_agree(pomgpu_ctx *c, int agreed) {
  if (!c) return POMGPU_EINVAL;
  pomgpu_transport &T = c->tp;
  if (!T.on) return pomgpu_fail(c, POMGPU_EINVAL, "transport_side_agree: set a transport first");
  if (c->wide.on) return pomgpu_fail(c, POMGPU_EINVAL, "transport_side_agree: before pomgpu_set_wide_external");
  if (agreed <= 0) { T.side_agreed = 0; T.wr_side = 0; return POMGPU_OK; }
  // "yes" can only come from the host's reduction over all ranks -- which included this rank's own answer
  const int mine = pomgpu_transport_side_capable(c);
  if (mine < 1 || agreed > mine) return pomgpu_fail(c, POMGPU_EINVAL, "transport_side_agree: %d cannot be the minimum over all ranks, this rank answered %d", agreed, mine);
  if (!T.fn) return POMGPU_OK;                                // RCCL: pomgpu_rccl_init has agreed already, nothing to raise
  T.side_agreed = 1;
  T.wr_side = agreed >= 2 ? 1 : 0;
  return POMGPU_OK;
}
// One message round on the SIDE stream: send2[d] -> neighbour d, recv2[d] <- neighbour d.  Every rank must issue its
// side-stream rounds in the same order (they do: one early gather and one wr exchange per internal step).
int pomgpu_tp_move_side(pomgpu_ctx *c, const size_t *scount, const size_t *rcount) {
  pomgpu_transport &T = c->tp;
  if (!pomgpu_tp_side_ok(c)) return pomgpu_fail(c, POMGPU_EINVAL, "transport: no side stream");
  T.rounds_side++;
  const int slot = c->prof_on ? pomgpu_prof_slot(c, "msg_round_side") : -1;
  if (slot >= 0) pomgpu_prof_pre(c);                          // c->cur is the side stream here
  if (T.fn) {                                                 // movers that stage through the host: hand the data over completed
    if (!T.fn_ordered) (void)hipStreamSynchronize(c->side);   // (a mover that enqueues on pomgpu_current_stream() orders itself)
    T.fn(T.user, T.send2, scount, T.recv2, rcount);
    if (slot >= 0) pomgpu_prof_post(c, slot);
    return POMGPU_OK;
  }
#ifndef POMGPU_EMU
  RcclComm *r = (RcclComm *)T.rccl;
  ncclResult_t e = g_rccl.GroupStart();
  for (int d = 0; d < 8 && e == ncclSuccess; d++)
    if (T.nbr[d] >= 0 && scount[d]) e = g_rccl.Send(T.send2[d], scount[d], ncclDouble, T.nbr[d], r->comm2, c->side);
  for (int d = 0; d < 8 && e == ncclSuccess; d++) {
    const int f = POMGPU_OPP[d];
    if (T.nbr[f] >= 0 && rcount[f]) e = g_rccl.Recv(T.recv2[f], rcount[f], ncclDouble, T.nbr[f], r->comm2, c->side);
  }
  const ncclResult_t e2 = g_rccl.GroupEnd();
  if (slot >= 0) pomgpu_prof_post(c, slot);
  if (e == ncclSuccess) e = e2;
  if (e != ncclSuccess) return pomgpu_fail(c, POMGPU_EHIP, "transport: RCCL round (side stream) failed: %s", g_rccl.GetErrorString(e));
  return POMGPU_OK;
#else
  return pomgpu_fail(c, POMGPU_ENODEV, "transport: no mover in the host build");
#endif
}

// One message round.  send[d] / recv[d]: device buffers (any, not only the staging buffers); counts in doubles.
int pomgpu_tp_move_ptr(pomgpu_ctx *c, const double *const *send, const size_t *scount, double *const *recv, const size_t *rcount) {
  pomgpu_transport &T = c->tp;
  if (!T.on) return pomgpu_fail(c, POMGPU_EINVAL, "transport: none set");
  T.rounds++;
  // profiling: a round is bracketed like a kernel ("msg_round": transfer time plus the wait for the neighbours)
  const int slot = c->prof_on ? pomgpu_prof_slot(c, "msg_round") : -1;
  if (slot >= 0) pomgpu_prof_pre(c);
  if (T.fn) {
    T.fn(T.user, send, scount, recv, rcount);
    if (slot >= 0) pomgpu_prof_post(c, slot);
    return POMGPU_OK;
  }
#ifndef POMGPU_EMU
  RcclComm *r = (RcclComm *)T.rccl;
  if (!r) return pomgpu_fail(c, POMGPU_EINVAL, "transport: no mover");
  ncclResult_t e = g_rccl.GroupStart();
  // Messages between one pair of ranks are matched in the order they are posted.  A pair normally meets in one
  // direction only; where a rank is its own neighbour in two directions (a periodic single-rank test) the
  // receives must be posted in the order of the directions they were SENT towards: d ascending on the sending
  // side is POMGPU_OPP[d] ascending in d on the receiving side.
  for (int d = 0; d < 8 && e == ncclSuccess; d++)
    if (T.nbr[d] >= 0 && scount[d]) e = g_rccl.Send(send[d], scount[d], ncclDouble, T.nbr[d], r->comm, c->stream);
  for (int d = 0; d < 8 && e == ncclSuccess; d++) {
    const int f = POMGPU_OPP[d];
    if (T.nbr[f] >= 0 && rcount[f]) e = g_rccl.Recv(recv[f], rcount[f], ncclDouble, T.nbr[f], r->comm, c->stream);
  }
  const ncclResult_t e2 = g_rccl.GroupEnd();
  if (slot >= 0) pomgpu_prof_post(c, slot);
  if (e == ncclSuccess) e = e2;
  if (e != ncclSuccess) return pomgpu_fail(c, POMGPU_EHIP, "transport: RCCL round failed: %s", g_rccl.GetErrorString(e));
  return POMGPU_OK;
#else
  return pomgpu_fail(c, POMGPU_ENODEV, "transport: no mover in the host build");
#endif
}
int pomgpu_tp_move(pomgpu_ctx *c, const size_t *scount, const size_t *rcount) {
  return pomgpu_tp_move_ptr(c, c->tp.send, scount, c->tp.recv, rcount);
}

// Can this process open librccl and find every entry point the transport uses?  Touches no GPU and no other rank:
// hosts all-gather the answer BEFORE anyone enters the collective ncclCommInitRank, so that no rank waits there for
// one that could not even load the library.
extern "C" int pomgpu_rccl_available(const char *librccl_path) {
#ifndef POMGPU_EMU
  return rccl_load(librccl_path) ? POMGPU_ENODEV : POMGPU_OK;
#else
  (void)librccl_path;
  return POMGPU_ENODEV;
#endif
}

extern "C" int pomgpu_rccl_unique_id(void *id128, const char *librccl_path) {
#ifndef POMGPU_EMU
  if (!id128) return POMGPU_EINVAL;
  if (rccl_load(librccl_path)) return POMGPU_ENODEV;
  ncclUniqueId id;
  const ncclResult_t e = g_rccl.GetUniqueId(&id);
  if (e != ncclSuccess) { fprintf(stderr, "pomgpu: ncclGetUniqueId: %s\n", g_rccl.GetErrorString(e)); return POMGPU_EHIP; }
  static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, sizeof id);
  return POMGPU_OK;
#else
  (void)id128; (void)librccl_path;
  return POMGPU_ENODEV;
#endif
}

// common part of pomgpu_set_transport / pomgpu_rccl_init: neighbour table, staging buffers for the ordinary
// exchange points (up to 8 arrays x kb levels of one edge line; corners one cell) and baropg_mcc's order messages
int pomgpu_tp_setup(pomgpu_ctx *c, const int *nbr8) {
  pomgpu_transport &T = c->tp;
  const KP &P = c->P;
  if (!nbr8) return pomgpu_fail(c, POMGPU_EINVAL, "transport: no neighbour table");
  if ((nbr8[0] < 0) != (P.W != 0) || (nbr8[1] < 0) != (P.E != 0) || (nbr8[2] < 0) != (P.S != 0) || (nbr8[3] < 0) != (P.N != 0))
    return pomgpu_fail(c, POMGPU_EINVAL, "transport: W E S N neighbours disagree with the tile's pomgpu_dims");
  for (int d = 0; d < 8; d++) T.nbr[d] = nbr8[d];
  size_t need[8];
  const size_t len[8] = {(size_t)P.jm, (size_t)P.jm, (size_t)P.im, (size_t)P.im, 1, 1, 1, 1};
  for (int d = 0; d < 8; d++) need[d] = 8 * (size_t)P.kb * len[d];
  const size_t ord[2] = {(size_t)(P.kb + 1) * P.jml, (size_t)(P.kb + 1) * P.iml};
  if (need[0] < ord[0]) need[0] = ord[0];
  if (need[1] < ord[0]) need[1] = ord[0];
  if (need[2] < ord[1]) need[2] = ord[1];
  if (need[3] < ord[1]) need[3] = ord[1];
  T.on = 1;
  T.rounds = 0;
  T.rounds_side = 0;
  return pomgpu_tp_reserve(c, need);   // side_agreed / wr_side: set by pomgpu_tp_rccl before this, 0 after pomgpu_tp_free otherwise
}

// how many ranks RCCL itself says the communicator has (ncclCommCount): 0 without the RCCL transport, -1 when the library
// cannot tell.  bench.py puts it on its line: a scaling number then shows that RCCL connected N ranks.
extern "C" int pomgpu_rccl_nranks(pomgpu_ctx *c) {
#ifndef POMGPU_EMU
  if (!c || !c->tp.on || !c->tp.rccl) return 0;
  RcclComm *r = (RcclComm *)c->tp.rccl;
  int n = -1;
  if (!g_rccl.CommCount || g_rccl.CommCount(r->comm, &n) != ncclSuccess) return -1;
  return n;
#else
  (void)c;
  return 0;
#endif
}
extern "C" long pomgpu_exchange_rounds(pomgpu_ctx *c) { return c ? c->tp.rounds : 0; }
extern "C" long pomgpu_exchange_rounds_side(pomgpu_ctx *c) { return c ? c->tp.rounds_side : 0; }

int pomgpu_tp_rccl(pomgpu_ctx *c, const void *id128, int rank, int nranks, const char *librccl_path) {
#ifndef POMGPU_EMU
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return pomgpu_fail(c, POMGPU_EINVAL, "rccl_init: bad rank / size");
  if (rccl_load(librccl_path)) return pomgpu_fail(c, POMGPU_ENODEV, "rccl_init: librccl could not be opened");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  RcclComm *r = new RcclComm();
  r->rank = rank; r->nranks = nranks; r->comm = NULL; r->comm2 = NULL;
  const ncclResult_t e = g_rccl.CommInitRank(&r->comm, nranks, id, rank);
  if (e != ncclSuccess) {
    delete r;
    return pomgpu_fail(c, POMGPU_EHIP, "rccl_init: ncclCommInitRank: %s", g_rccl.GetErrorString(e));
  }
  // Everything below that is collective depends on answers every rank gives for itself: its switches, whether it can make a
  // second stream, whether its library has ncclCommSplit.  So the answers are reduced over the communicator FIRST (one
  // all-reduce that every rank reaches whatever it answers) and only what ALL ranks agreed on is then done by all of them:
  //   v[0] may a second communicator be split off (min)      v[1] may wr go to the side stream (min)
  //   v[2], v[3] the digest of the switches every rank must share and its negative (min of both: equal iff all ranks agree)
  // A rank on which an all-reduce itself fails leaves with an error; its partners' deadline (bench.py, the host's own) ends
  // them -- they cannot be told.
  auto allmin = [&](int *v, int n) -> const char * {          // NULL = done, v holds the minimum over the ranks
    int *d = NULL;
    bool ok = hipMalloc((void **)&d, 2 * n * sizeof(int)) == hipSuccess;
    ok = ok && hipMemcpyAsync(d, v, n * sizeof(int), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    ncclResult_t ea = ncclSuccess;
    if (ok) ea = g_rccl.AllReduce(d, d + n, n, ncclInt32, ncclMin, r->comm, c->stream);
    ok = ok && ea == ncclSuccess && hipMemcpyAsync(v, d + n, n * sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
    (void)hipFree(d);
    return ok ? NULL : (ea != ncclSuccess ? g_rccl.GetErrorString(ea) : "HIP");
  };
  auto give_up = [&](const char *what, const char *why) {
    if (r->comm2) (void)g_rccl.CommDestroy(r->comm2);
    (void)g_rccl.CommDestroy(r->comm);
    delete r;
    return pomgpu_fail(c, POMGPU_EHIP, "rccl_init: %s: %s", what, why);
  };
  const int digest = (int)pomgpu_switches_collective_digest(c->sw);
  int v[4] = {(g_rccl.CommSplit && side_capable_local(c)) ? 1 : 0, SW(c, WR_MAIN) ? 0 : 1, digest, -digest};
  const int mine_split = v[0];
  if (const char *why = allmin(v, 4)) return give_up("the ranks' agreement on side-stream rounds failed", why);
  if (v[2] != -v[3])                                          // every rank sees the same two minima: all of them refuse together
    return give_up("the ranks were started with different POMGPU_* switch sets",
                   "ADVCT_SPLIT ADVQ_EXCHANGE PROD_FULL QFILTER_SPLIT UV_FULL_EXCHANGE NO_OVERLAP NO_SIDE_COMM WR_MAIN WIDE_W WIDE_FULL EXT_SPLIT "
                   "ADVAVE_SEPARATE EDGE_SPLIT WR_NODEFER RIM_MAIN RIM_RESULTS_MAIN ADVT2_SINGLE SUM2D_OFF choose which message rounds exist: give every rank the same environment");
  int agreed[2] = {v[0], v[1]};
  if (agreed[0]) {
    // the side stream's communicator: the same ranks, split off the first -- collective, entered by ALL ranks or by none
    const ncclResult_t e2 = g_rccl.CommSplit(r->comm, 0, rank, &r->comm2, NULL);
    if (e2 != ncclSuccess) { r->comm2 = NULL; fprintf(stderr, "pomgpu: ncclCommSplit: %s -- message rounds stay on one stream\n", g_rccl.GetErrorString(e2)); }
    // tests (a single rank that is its own neighbour, tests/gpu_rccl_self.py): this rank behaves as if its split had failed
    if (SW(c, TEST_SPLIT_FAIL_RANK) && (int)SWV(c, TEST_SPLIT_FAIL_RANK) == rank) {
      fprintf(stderr, "pomgpu: POMGPU_TEST_SPLIT_FAIL_RANK=%d is set -- rank %d reports a failed ncclCommSplit (test hook)\n", rank, rank);
      if (r->comm2) { (void)g_rccl.CommDestroy(r->comm2); r->comm2 = NULL; }
    }
    int got[1] = {r->comm2 != NULL ? 1 : 0};
    if (const char *why = allmin(got, 1)) return give_up("the ranks' agreement on the second communicator failed", why);
    agreed[0] = got[0];
  }
  if (!agreed[0] && r->comm2) { (void)g_rccl.CommDestroy(r->comm2); r->comm2 = NULL; }
  if (!agreed[0] && mine_split && rank == 0) fprintf(stderr, "pomgpu: a rank cannot serve side-stream rounds -- every rank keeps its message rounds on one stream\n");
  c->tp.side_agreed = agreed[0];
  c->tp.wr_side = agreed[0] && agreed[1];
  c->tp.rccl = r;
  c->tp.fn = NULL;
  c->tp.user = NULL;
  return POMGPU_OK;
#else
  (void)id128; (void)rank; (void)nranks; (void)librccl_path;
  return pomgpu_fail(c, POMGPU_ENODEV, "rccl_init: the host build has no RCCL");
#endif
}
