#!/bin/bash
# Where the register allocator spills in td_kernel<FUSED>: source lines of the scratch spill / reload instructions (no GPU needed).
# usage: bash tools/spill_sites.sh [extra hipcc flags, e.g. -DSCG_EO_MFMA]
cd "$(dirname "$0")/../skill-chaining-with-graphs_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -gline-tables-only -S --cuda-device-only "$@" -o /tmp/scg_td_g.s scg_kernels.hip 2>/dev/null || exit 1
awk '/^_Z9td_kernelILi0EEv8StepArgs:/,/s_endpgm/' /tmp/scg_td_g.s > /tmp/scg_td0g.s
python3 - <<'PY'
import re, collections
cur = None
st = collections.Counter(); ld = collections.Counter(); lane = 0
for l in open('/tmp/scg_td0g.s'):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if 'Folded Spill' in l: st[cur] += 1
    if 'Folded Reload' in l: ld[cur] += 1
    if re.search(r'v_(writelane|readlane)_b32', l): lane += 1
print("scratch spills  (file, line): n ", sorted(st.items(), key=lambda x: -x[1])[:20])
print("scratch reloads (file, line): n ", sorted(ld.items(), key=lambda x: -x[1])[:20])
print("SGPR lane moves:", lane)
PY
grep -n '\.file' /tmp/scg_td_g.s | grep -E 'scg_|\.file\s+[0-9]+ "\."' | head -5
