#!/bin/bash
# build libneo_mpc.so from a (patched) copy of the working tree's sources: tools/_build/libneo_mpc_<name>.so
# usage: bash tools/build_tree.sh <name> [sed expression on neo_mpc_kernels.hip ...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/neo_tree_$NAME
rm -rf $D && mkdir -p $D/neo_mpc_planner2_amd $ROOT/tools/_build
cp -r $ROOT/include $D/; cp -r $ROOT/neo_mpc_planner2_amd/csrc $D/neo_mpc_planner2_amd/
for e in "$@"; do sed -i "$e" $D/neo_mpc_planner2_amd/csrc/neo_mpc_kernels.hip; done
make -C $D/neo_mpc_planner2_amd/csrc -j4 > $D/make.log 2>&1 || { tail -20 $D/make.log; exit 1; }
cp $D/neo_mpc_planner2_amd/libneo_mpc.so $ROOT/tools/_build/libneo_mpc_$NAME.so
