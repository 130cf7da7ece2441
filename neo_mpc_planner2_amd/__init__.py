"""MI355X-native batched MPC solver: drop-in for the optimisation inner loop of
neobotix/neo_mpc_planner2 (SciPy SLSQP `minimize` + objective in
neo_mpc_planner2/mpc_optimization_server.py:204-269, 363-364).

The compute path is the HIP library `libneo_mpc.so` (see `include/neo_mpc.h`);
importing this package does not load it, constructing a solver does and fails
loudly when it is missing.
"""
__version__ = "0.1.0"
