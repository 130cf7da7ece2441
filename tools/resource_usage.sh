#!/bin/bash
# per-variant register / spill / scratch summary of K1 (both translation units)   usage: bash tools/resource_usage.sh [filter]
cd "$(dirname "$0")/../neo_mpc_planner2_amd/csrc"
for tu in neo_mpc_kernels.hip neo_mpc_riccati.hip; do
  extra=""; [ $tu = neo_mpc_riccati.hip ] && extra="-fno-slp-vectorize -mllvm -disable-machine-licm"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed $extra \
    -Rpass-analysis=kernel-resource-usage -x hip -c $tu -o /dev/null 2>&1 | python3 -c "
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
for k,v in rows.items():
    if 'k_solve' not in k: continue
    name=re.sub(r'.*k_solve_routedI','k_solve_routed<',k) if 'k_solve_routed' in k else re.sub(r'.*k_solveI','k_solve<',k)
    name=name.replace('EEEvNS_9SolveArgsE','>').replace('ELi',',').replace('ELb',',b').replace('Li','').replace('Lb','b')
    if ('$tu'=='neo_mpc_riccati.hip') != (',0,2,' in name or 'routed' in name): continue
    print('%-34s VGPR %3d SGPR %3d sgpr-spill %3d vgpr-spill %3d scratch %3d occ %d' % (name, v.get('VGPRs',0), v.get('SGPRs',0), v.get('SGPRs Spill',0), v.get('VGPRs Spill',0), v.get('ScratchSize',0), v.get('Occupancy',0)))
" | grep "${1:-.}"
done
