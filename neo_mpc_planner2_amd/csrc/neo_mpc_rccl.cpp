// neo_mpc_rccl.cpp -- the one exchange step of a multi-GPU fleet through the C-ABI: an all-gather of the packed
// (vx, vy, omega) commands over RCCL (xGMI), and the one-off broadcast of a raw costmap.
//
// The reference has no collective (a single robot, a ROS2 service hop, src/NeoMpcPlanner.cpp:248-250); this is
// the fleet caller's counterpart of `return out->output_vel` (cpp:251-254) when the instances of one tick are
// sharded over the GPUs of a node.  RCCL is bound at run time (dlopen): libneo_mpc.so loads -- and the
// single-GPU plugin path works -- on hosts without it; the entry points then return NEO_MPC_ERR_UNSUPPORTED.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "neo_mpc_device.h"

int neo_mpc_set_error(int code, const char* fmt, ...);   // neo_mpc_capi.cpp

namespace {

// the handful of RCCL entry points this file needs (signatures: rccl.h of ROCm 7)
typedef void* comm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*CommCount)(comm_t, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  char why[256] = "";   // why it is not: the dlopen error, captured where it happened (dlerror() reads once, per thread)
};
constexpr int kNcclUint8 = 1, kNcclDouble = 8;   // ncclDataType_t

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // NEO_MPC_RCCL_LIBRARY=<path>: the library to bind instead of the system's librccl.so (read here, once) -- a site build
    // of RCCL, or the tests' stand-in (tests/standin_rccl: several logical ranks on the one device a build box has)
    const char* forced = getenv("NEO_MPC_RCCL_LIBRARY");
    for (const char* name : {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (!name || !*name) continue;
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      const char* e = dlerror();
      if (e) snprintf(r.why, sizeof(r.why), "%s", e);
      if (name == forced) return;   // (the caller asked for THAT library: no silent fallback to another)
    }
    if (!r.lib) return;
#define NEO_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
    NEO_SYM(CommInitAll, "ncclCommInitAll"); NEO_SYM(CommDestroy, "ncclCommDestroy"); NEO_SYM(CommCount, "ncclCommCount");
    NEO_SYM(AllGather, "ncclAllGather"); NEO_SYM(Broadcast, "ncclBroadcast"); NEO_SYM(GroupStart, "ncclGroupStart");
    NEO_SYM(GroupEnd, "ncclGroupEnd"); NEO_SYM(GetErrorString, "ncclGetErrorString");
#undef NEO_SYM
    r.ok = r.CommInitAll && r.CommDestroy && r.CommCount && r.AllGather && r.Broadcast && r.GroupStart && r.GroupEnd &&
           r.GetErrorString;
    if (!r.ok) snprintf(r.why, sizeof(r.why), "librccl.so lacks one of the nccl* entry points this library binds");
  });
  return r;
}

int need_rccl() {
  if (!rccl().ok) return neo_mpc_set_error(NEO_MPC_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: %s", rccl().why);
  return NEO_MPC_OK;
}

int check(int rc, const char* what) {
  if (rc != 0) return neo_mpc_set_error(NEO_MPC_ERR_DEVICE, "%s failed: %s", what, rccl().GetErrorString(rc));
  return NEO_MPC_OK;
}

}  // namespace

extern "C" {

int neo_mpc_rccl_available(void) { return need_rccl() == NEO_MPC_OK ? 1 : 0; }   // (0: neo_mpc_last_error() says why)

int neo_mpc_comm_init_all(int ndev, const int* devices, void** comms_out) {
  if (ndev <= 0 || !comms_out) return neo_mpc_set_error(NEO_MPC_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  int rc = need_rccl();
  if (rc) return rc;
  return check(rccl().CommInitAll(comms_out, ndev, devices), "ncclCommInitAll");
}

int neo_mpc_comm_destroy(void* comm) {
  if (!comm) return NEO_MPC_OK;
  int rc = need_rccl();
  if (rc) return rc;
  return check(rccl().CommDestroy(comm), "ncclCommDestroy");
}

int neo_mpc_group_start(void) {
  int rc = need_rccl();
  return rc ? rc : check(rccl().GroupStart(), "ncclGroupStart");
}

int neo_mpc_group_end(void) {
  int rc = need_rccl();
  return rc ? rc : check(rccl().GroupEnd(), "ncclGroupEnd");
}

int neo_mpc_allgather_velocities(const double* d_local, double* d_all, size_t count, void* comm, void* stream) {
  if (!d_local || !d_all || !comm) return neo_mpc_set_error(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  int rc = need_rccl();
  if (rc) return rc;
  return check(rccl().AllGather(d_local, d_all, 3 * count, kNcclDouble, comm, (hipStream_t)stream), "ncclAllGather");
}

int neo_mpc_broadcast_costmap(uint8_t* d_cells, size_t bytes, int root, void* comm, void* stream) {
  if (!d_cells || !comm) return neo_mpc_set_error(NEO_MPC_ERR_INVALID_ARGUMENT, "null argument");
  int rc = need_rccl();
  if (rc) return rc;
  return check(rccl().Broadcast(d_cells, d_cells, bytes, kNcclUint8, root, comm, (hipStream_t)stream), "ncclBroadcast");
}

}  // extern "C"
