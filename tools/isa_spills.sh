#!/bin/bash
# Where the spill reloads of the TRAIN 0 multi-tick kernels sit (round 6: every reload is a scratch load + s_waitcnt vmcnt(0), i.e. it also waits
# for every weight-ring load / observation-row store the wave has in flight -- DESIGN.md 5.12):   tools/isa_spills.sh [0|1] [extra hipcc flags]
#   unit 0 = k_run<512, fixed, dueling, TRAIN 0> (configs[3]), unit 1 = k_run<512, fixed, kKindAll, TRAIN 0> (configs[4])
cd "$(dirname "$0")/.." || exit 1
u=${1:-0}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --cuda-device-only -DRL_RUN_UNIT=$u "$@" -S reinlife_amd/csrc/rl_run.hip -o /tmp/isa_u$u.s 2>/dev/null || exit 1
E=$(grep -n "^\.Lfunc_end0:" /tmp/isa_u$u.s | cut -d: -f1)
awk -v e=$E 'NR<e' /tmp/isa_u$u.s > /tmp/isa_k$u.s
grep -E "^\s*; (ScratchSize|NumVgprs|VGPRs Spill|SGPRs Spill)|\.vgpr_spill_count|\.private_segment_fixed_size" /tmp/isa_u$u.s | head -4
echo "first/last MFMA line: $(grep -n v_mfma /tmp/isa_k$u.s | head -1 | cut -d: -f1) / $(grep -n v_mfma /tmp/isa_k$u.s | tail -1 | cut -d: -f1)   tick loop header: $(grep -n 'Loop Header: Depth=1' /tmp/isa_k$u.s | head -1 | cut -d: -f1)"
awk '/s_barrier/{b++} /v_mfma/{m++} /scratch_load/{printf "reload at line %d  (after barrier #%d, %d MFMAs before it): %s\n", NR, b, m, $0}' /tmp/isa_k$u.s
