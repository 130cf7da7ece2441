#!/bin/bash
# build libneo_mpc.so of another revision for a same-box A/B (tools/ab_many.sh): tools/_build/libneo_mpc_<name>.so
# usage: bash tools/build_variant.sh <git revision> <name>
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/neo_variant_$NAME
rm -rf $D && mkdir -p $D $ROOT/tools/_build
git -C $ROOT archive $REV neo_mpc_planner2_amd/csrc include | tar -x -C $D
make -C $D/neo_mpc_planner2_amd/csrc -j4 > $D/make.log 2>&1 || { tail -20 $D/make.log; exit 1; }
cp $D/neo_mpc_planner2_amd/libneo_mpc.so $ROOT/tools/_build/libneo_mpc_$NAME.so
ls -la $ROOT/tools/_build/libneo_mpc_$NAME.so
