"""Where does the dispatcher put wave w of the step kernel's 16-wave workgroup? Every design note since round 3 ASSUMES SIMD = w % 4.
The light timing build (`make -C .../csrc lite`) stores HW_REG_HW_ID of every wave of every block of the last launch (slots 34..41);
this prints the wave -> SIMD table over all blocks.   python tools/wave_placement.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from skill_chaining_with_graphs_amd import _lib
_lib.LIB_PATH = os.environ.get("SCG_LITE_LIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "libscg_hip_lite.so")
import numpy as np, torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
n = 65536
agent = SkillChainingAgent(bench.MAP, n, 5, seed=0, **bench.HP)
agent.clf.copy_(torch.as_tensor(bench.chain_discs(agent.map, 5)))
for k in range(1, 6): agent.enable_option(k)
agent.init_weights(std=1e-3); agent.domain.reset_random(seed=1000)
lib, ctx = agent.ctx.lib, agent.ctx._ctx
lib.scg_diag_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
nblk = n // lib.scg_block_envs()
for rep in range(3):
    lib.scg_diag_stamps(ctx, None, 1)
    for _ in range(20): agent.step_batch()
    torch.cuda.synchronize()
    out = np.zeros((nblk, 48), np.uint64)
    lib.scg_diag_stamps(ctx, out.ctypes.data_as(C.c_void_p), 0)
    hw = out[:, 34:42].copy().view(np.uint32).reshape(nblk, 16)
    simd, wid, cu, se = (hw >> 4) & 3, hw & 15, (hw >> 8) & 15, (hw >> 13) & 7
    print(f"launch set {rep}: wave -> SIMD_ID, blocks with that placement")
    pats, cnt = np.unique(simd, axis=0, return_counts=True)
    for p, c in sorted(zip(pats.tolist(), cnt.tolist()), key=lambda t: -t[1])[:8]:
        print("   ", "".join(str(v) for v in p), f" x {c}   per-SIMD wave counts {np.bincount(p, minlength=4).tolist()}")
    print("    is w % 4 everywhere:", bool((simd == (np.arange(16) % 4)[None, :]).all()), " is w // 4 everywhere:", bool((simd == (np.arange(16) // 4)[None, :]).all()))
    print("    SIMDs of waves 12, 13, 14 (the no-op's U2 waves), first 8 blocks:", simd[:8, 12:15].tolist())
    print("    WAVE_ID of waves 0..15, block 0:", wid[0].tolist(), " CU", cu[0, 0], "SE", se[0, 0])
