// neo_mpc_riccati.hip -- the Riccati variants of K1 (k_solve<*, 0, 2, *>) as a translation unit of their own:
// the same source as neo_mpc_kernels.hip, compiled with -fno-slp-vectorize (see launch_solve_riccati there).
// Part of libneo_mpc.so.
#define NEO_MPC_TU_RICCATI
#include "neo_mpc_kernels.hip"
