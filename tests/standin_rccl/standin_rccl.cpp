// standin_rccl.cpp -- TEST INFRASTRUCTURE, not RCCL: a stand-in librccl.so that implements the eight nccl* entry points
// libneo_mpc.so binds at run time (neo_mpc_rccl.cpp) with plain HIP copies on ONE device, so that the multi-rank path of the
// C-ABI -- neo_mpc_comm_init_all with more than one rank, the group bracketing, per-rank communicators and streams, the
// offsets and the layout of the gathered buffer (examples/fleet_allgather.cpp) -- executes on the hardware there is: the
// build box has one GPU, RCCL refuses two ranks on one device and the lease refuses to partition it.  What it proves is the
// CALLER's code, "functional, not RCCL": nothing about xGMI, nothing about RCCL's own behaviour.
// Selected with NEO_MPC_RCCL_LIBRARY=<path of this library> (read once, where neo_mpc_rccl.cpp opens librccl.so).
//
// Semantics kept: a communicator per rank from ncclCommInitAll; collectives issued between ncclGroupStart and ncclGroupEnd
// take effect at ncclGroupEnd; a collective is ordered behind everything queued on EVERY participating rank's stream and
// in front of what is queued on them afterwards (events + stream waits); ncclAllGather puts rank s's `count` elements at
// offset s * count of every rank's receive buffer; ncclBroadcast copies the root's buffer to every other rank's.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
struct Group;
struct Comm { Group* group; int rank; };
struct Pending { int kind; const void* send; void* recv; size_t bytes; int root; Comm* comm; hipStream_t stream; };
struct Group {
  int nranks = 0;
  std::vector<Comm*> comms;
  std::vector<Pending> pending;   // one collective per rank between GroupStart and GroupEnd
};
std::mutex g_lock;
int g_depth = 0;
std::vector<Group*> g_open;   // groups with pending calls
size_t type_size(int t) { return t == 1 ? 1 : t == 8 ? 8 : t == 7 ? 4 : t == 2 ? 4 : t == 9 ? 2 : 0; }   // uint8, double, float, int32, half

int flush(Group* g) {
  if (g->pending.empty()) return 0;
  if ((int)g->pending.size() != g->nranks) { g->pending.clear(); return 5; }   // ncclInvalidUsage: every rank must call
  const int n = g->nranks;
  std::vector<const Pending*> by_rank(n, nullptr);
  for (const Pending& p : g->pending) by_rank[p.comm->rank] = &p;
  for (int r = 0; r < n; ++r) if (!by_rank[r] || by_rank[r]->kind != by_rank[0]->kind || by_rank[r]->bytes != by_rank[0]->bytes) { g->pending.clear(); return 5; }
  // behind everything queued on every rank's stream ...
  std::vector<hipEvent_t> ready(n);
  for (int r = 0; r < n; ++r) {
    if (hipEventCreateWithFlags(&ready[r], hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(ready[r], by_rank[r]->stream) != hipSuccess) return 1;
  }
  for (int r = 0; r < n; ++r)
    for (int s = 0; s < n; ++s)
      if (s != r && hipStreamWaitEvent(by_rank[r]->stream, ready[s], 0) != hipSuccess) return 1;
  // ... the copies, each rank's on its own stream ...
  for (int r = 0; r < n; ++r) {
    const Pending& p = *by_rank[r];
    if (p.kind == 0) {          // all-gather
      for (int s = 0; s < n; ++s)
        if (hipMemcpyAsync((char*)p.recv + (size_t)s * p.bytes, by_rank[s]->send, p.bytes, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) return 1;
    } else if (r != p.root) {   // broadcast
      if (hipMemcpyAsync(p.recv, by_rank[p.root]->send, p.bytes, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) return 1;
    }
  }
  // ... and in front of what the ranks queue afterwards: nobody overwrites a send buffer another rank still reads
  std::vector<hipEvent_t> done(n);
  for (int r = 0; r < n; ++r) {
    if (hipEventCreateWithFlags(&done[r], hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(done[r], by_rank[r]->stream) != hipSuccess) return 1;
  }
  for (int r = 0; r < n; ++r)
    for (int s = 0; s < n; ++s)
      if (s != r && hipStreamWaitEvent(by_rank[r]->stream, done[s], 0) != hipSuccess) return 1;
  for (int r = 0; r < n; ++r) { (void)hipEventDestroy(ready[r]); (void)hipEventDestroy(done[r]); }
  g->pending.clear();
  return 0;
}

int enqueue(Pending p) {
  std::lock_guard<std::mutex> hold(g_lock);
  Group* g = p.comm->group;
  g->pending.push_back(p);
  bool open = false;
  for (Group* o : g_open) open = open || o == g;
  if (!open) g_open.push_back(g);
  if (g_depth > 0) return 0;
  // outside a group bracket only a communicator of one rank can complete on its own
  g_open.clear();
  return flush(g);
}
}  // namespace

extern "C" {
int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
  (void)devlist;   // (the real RCCL refuses duplicate devices: this stand-in is what runs several ranks on one)
  if (!comms || ndev < 1) return 4;
  Group* g = new Group;
  g->nranks = ndev;
  for (int r = 0; r < ndev; ++r) { Comm* c = new Comm{g, r}; g->comms.push_back(c); comms[r] = c; }
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  Group* g = c->group;
  g->comms[c->rank] = nullptr;
  bool any = false;
  for (Comm* o : g->comms) any = any || o != nullptr;
  delete c;
  if (!any) delete g;
  return 0;
}
int ncclCommCount(void* comm, int* count) { if (!comm || !count) return 4; *count = ((Comm*)comm)->group->nranks; return 0; }
int ncclGroupStart() { std::lock_guard<std::mutex> hold(g_lock); ++g_depth; return 0; }
int ncclGroupEnd() {
  std::lock_guard<std::mutex> hold(g_lock);
  if (g_depth <= 0) return 5;
  if (--g_depth > 0) return 0;
  int rc = 0;
  for (Group* g : g_open) { const int r = flush(g); if (r) rc = r; }
  g_open.clear();
  return rc;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  const size_t ts = type_size(dtype);
  if (!send || !recv || !comm || !ts) return 4;
  return enqueue(Pending{0, send, recv, count * ts, 0, (Comm*)comm, stream});
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream) {
  const size_t ts = type_size(dtype);
  if (!send || !recv || !comm || !ts || root < 0 || root >= ((Comm*)comm)->group->nranks) return 4;
  return enqueue(Pending{1, send, recv, count * ts, root, (Comm*)comm, stream});
}
const char* ncclGetErrorString(int rc) {
  return rc == 0 ? "no error" : rc == 1 ? "stand-in: a HIP call failed" : rc == 4 ? "stand-in: invalid argument"
                                                                                 : "stand-in: invalid usage (every rank of a communicator calls once per group)";
}
}
