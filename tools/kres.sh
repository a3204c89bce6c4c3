#!/bin/bash
# Resource usage (VGPRs, spills, scratch) of the kernels of one translation unit:  tools/kres.sh rl_run.hip [filter-regex] [extra hipcc flags]
cd "$(dirname "$0")/.." || exit 1
src=reinlife_amd/csrc/${1:-rl_run.hip}; filt=${2:-k_run}; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /tmp/kres.o 2>&1 \
  | grep -E "Function Name|VGPRs:|Spill|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
  | awk '/Function Name/ {name=$3} /VGPRs:/ {v=$2} /ScratchSize/ {sc=$3} /SGPRs Spill/ {ss=$3} /VGPRs Spill/ {print name, "vgpr", v, "scratch", sc, "sgpr_spill", ss, "vgpr_spill", $3}' | c++filt | grep -E "$filt"
