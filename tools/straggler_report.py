"""Which workgroups of the step kernel are the slow ones? The launch lasts as long as its slowest workgroup (one per CU, one round).
Light timing build: every workgroup's start and end of the LAST launch on the 100 MHz clock; the env order of that launch is re-made on
the host from the option ids the step started with (SPEC §5: key = the option an env runs, 0 for the root and for envs staying out;
stable in the env id), so every block's wall time can be set beside what it held.   python tools/straggler_report.py [--launches K]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
ap = argparse.ArgumentParser(); ap.add_argument("--launches", type=int, default=24)
ap.add_argument("--rotate", type=int, default=0, help="the library was built with -DSCG_DIAG_ROTATE=R (make lite_rot: 100): workgroup p holds block (p + R) %% grid")
ap.add_argument("--pad-kb", type=int, default=0, help="allocate this much device memory BEFORE the agent's buffers (moves every buffer: does the slow set follow the addresses?)")
ap.add_argument("--p-internal", action="store_true", help="the library is the SCG_STAMPS_LITE=2 build (make lite_p): env wave 0's boundaries lie INSIDE phase P")
ap.add_argument("--options", type=int, default=5); ap.add_argument("--seed", type=int, default=1000)
args = ap.parse_args()
from skill_chaining_with_graphs_amd import _lib
_lib.LIB_PATH = os.environ.get("SCG_LITE_LIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "libscg_hip_lite.so")
import numpy as np, torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
n, nopt = 65536, args.options
_pad = torch.empty(args.pad_kb * 1024, dtype=torch.uint8, device="cuda") if args.pad_kb else None
agent = SkillChainingAgent(bench.MAP, n, nopt, seed=0, **bench.HP)
if nopt: agent.clf.copy_(torch.as_tensor(bench.chain_discs(agent.map, nopt)))
for k in range(1, nopt + 1): agent.enable_option(k)
agent.init_weights(std=1e-3); agent.domain.reset_random(seed=args.seed)
for _ in range(300): agent.step_batch()
torch.cuda.synchronize()
lib, ctx = agent.ctx.lib, agent.ctx._ctx
lib.scg_diag_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
B = lib.scg_block_envs(); nblk = n // B
lib.scg_diag_stamps(ctx, None, 1)
def env_order(key):
    """SPEC §5's CHUNKED layout restated on the host (csrc/scg_kernels.hip order_layout / order_posk / order_pos0): every block starts with at most c envs of
    ONE option's run and is filled with root-keyed envs behind them; ranks inside a key are by env id. Returns order[pos] = env."""
    nk = 7
    tot = np.bincount(key, minlength=nk)
    S, Rn, Bf = int(tot[1:].sum()), int((tot[1:] > 0).sum()), n // B
    c = B
    if Bf > Rn and S > 0: c = min(B, -(-S // (Bf - Rn)))
    g = B - c
    cnt = np.zeros(nk, int); start = np.zeros(nk, int); F = np.zeros(nk, int)
    U = Fs = 0
    for k in range(1, nk):
        cnt[k] = -(-int(tot[k]) // c); start[k] = U * B; F[k] = Fs
        U += cnt[k]; Fs += cnt[k] * B - int(tot[k])
    assert U * B <= n, "padded layout: not restated here"
    pos = np.empty(n, np.int64)
    for k in range(1, nk):
        e = np.nonzero(key == k)[0]; r = np.arange(len(e)); t = r // c
        pos[e] = start[k] + B * t + (r - t * c)
    e0 = np.nonzero(key == 0)[0]; r = np.arange(len(e0))
    p0 = U * B + (r - Fs)
    for k in range(1, nk):
        fills = cnt[k] * B - int(tot[k])
        m = (r >= F[k]) & (r < F[k] + fills)
        rp = r[m] - F[k]; nfull = cnt[k] - 1
        a = np.where((g > 0) & (rp < nfull * g), start[k] + B * (rp // max(g, 1)) + c + (rp % max(g, 1)),
                     start[k] + B * nfull + (int(tot[k]) - nfull * c) + (rp - nfull * g))
        p0[m] = a
    pos[e0] = p0
    order = np.empty(n, np.int64); order[pos] = np.arange(n)
    assert (np.sort(pos) == np.arange(n)).all()
    return order
rows = []; ticks = []; t0s = []; extra = []; groups = []
for it in range(args.launches):
    for _ in range(3): agent.step_batch()
    torch.cuda.synchronize()
    opt = agent.state.option_id.cpu().numpy().astype(np.int64)
    pre_xy = (agent.state.x.cpu().numpy().copy(), agent.state.y.cpu().numpy().copy()); pre_v = (agent.state.vx.cpu().numpy().copy(), agent.state.vy.cpu().numpy().copy())
    lib.scg_diag_stamps(ctx, None, 1)                     # the cycle stamps accumulate: zero them, so that they are this launch's alone
    agent.step_batch(); torch.cuda.synchronize()
    out = np.zeros((nblk, 48), np.uint64)
    lib.scg_diag_stamps(ctx, out.ctypes.data_as(C.c_void_p), 0)
    if args.rotate:                                        # stamps are stored per WORKGROUP: re-index them by the block the workgroup held
        out = out[(np.arange(nblk) - args.rotate) % nblk]
    dur = (out[:, 33].astype(np.int64) - out[:, 32].astype(np.int64)) / 100.0
    groups.append(out[:, 42].astype(np.float64))
    ticks.append(out[:, :8].astype(np.float64))           # env wave 0: [0] whole kernel, [1..7] the phase boundaries (s_memtime since entry)
    t0s.append((out[:, 32].astype(np.int64) - out[:, 32].astype(np.int64).min()) / 100.0)
    key = np.where(opt > 0, opt, 0)
    order = env_order(key)
    if it == args.launches - 1:                            # what the blocks of the last launch held, by position in the env order
        st = agent.state
        opt_after = st.option_id.cpu().numpy().astype(np.int64)
        vx, vy = pre_v
        feats = {"env id / 1000": order / 1000.0, "x": pre_xy[0][order], "y": pre_xy[1][order], "|v|": np.hypot(vx, vy)[order],
                 "staying out": (opt[order] < 0) * 1.0, "stays out of 5": (opt[order] == -5) * 1.0, "of 4": (opt[order] == -4) * 1.0, "of 1..3": ((opt[order] < 0) & (opt[order] > -4)) * 1.0,
                 "option changed": (opt_after[order] != opt[order]) * 1.0, "episode steps / 100": st.ep_steps.cpu().numpy()[order] / 100.0,
                 "action = no-op": (st.action.cpu().numpy()[order] == 4) * 1.0}
        print("last launch, means per 16 blocks of the env order (4096 envs each):")
        for nm, v in feats.items():
            print(f"  {nm:22s}" + " ".join(f"{x:7.2f}" for x in v.reshape(16, -1).mean(1)))
    ob, kb = opt[order].reshape(nblk, B), key[order].reshape(nblk, B)
    oa = agent.state.option_id.cpu().numpy().astype(np.int64)[order].reshape(nblk, B)
    epa = agent.state.ep_steps.cpu().numpy().astype(np.int64)[order].reshape(nblk, B)
    extra.append(np.stack([((ob <= 0) & (oa > 0)).sum(1), (epa <= 1).sum(1), ((ob < 0) & (oa == 0)).sum(1), ((ob == 0) & (oa < 0)).sum(1),
                           np.array([len(np.unique(order[b * B:(b + 1) * B] >> 8)) for b in range(nblk)])], 1))
    for b in range(nblk):
        ks = np.unique(kb[b])
        rows.append((dur[b], int((kb[b] > 0).sum()), int((ob[b] < 0).sum()), len(ks), int(ks.max()), len(np.unique(ob[b][ob[b] < 0])), b, it))
r = np.array(rows, dtype=np.float64)
d = r[:, 0]
print(f"{args.launches} launches x {nblk} blocks; block wall time us: mean {d.mean():.2f}  median {np.median(d):.2f}  p90 {np.percentile(d, 90):.2f}  p99 {np.percentile(d, 99):.2f}  max {d.max():.2f}")
per = r[:, 0].reshape(args.launches, nblk)
print(f"per launch: mean of block means {per.mean(1).mean():.2f} us, mean of block maxima {per.max(1).mean():.2f} us  (the launch pays the maximum: {100 * (per.max(1).mean() / per.mean(1).mean() - 1):.1f} % over the mean)")
def grp(name, m):
    if m.sum(): print(f"  {name:58s} {int(m.sum()):6d} blocks  mean {d[m].mean():6.2f}  p90 {np.percentile(d[m], 90):6.2f}  max {d[m].max():6.2f}")
inopt, stay, nk = r[:, 1], r[:, 2], r[:, 3]
for kk in range(0, nopt + 1):
    grp(f"blocks that start with option {kk}'s envs" if kk else "blocks of root-keyed envs only", (r[:, 4] == kk))
print("envs of the block's option per block: mean %.1f, max %d" % (inopt.mean(), inopt.max()))
top = np.argsort(d)[-16:][::-1]
print("slowest 16 (us, envs in an option, staying out, keys in block, option, launch, block):")
for i in top: print(f"   {d[i]:6.2f}  {int(r[i,1]):4d} {int(r[i,2]):4d} {int(r[i,3]):2d} {int(r[i,4]):2d}   launch {int(r[i,7]):2d} block {int(r[i,6]):3d}")
slow_each = per.argmax(1)
print("the slowest block of each launch:", slow_each.tolist())

tk = np.concatenate(ticks, 0); t0 = np.concatenate(t0s, 0)
G = np.concatenate(groups, 0)
print("pair groups of the block's physics (64 (env, edge) pairs each; seven waves take one each per round) -> blocks, mean wall us, share of the per-launch maxima:")
Gb = G.reshape(args.launches, nblk)
print("mean pair groups by position in the env order (16 bins of 16 blocks):", " ".join(f"{v:.2f}" for v in Gb.mean(0).reshape(16, -1).mean(1)))
print("share of launches with 8 groups or more, blocks 20..47:", " ".join(f"{(Gb[:, b] >= 8).mean():.2f}" for b in range(20, 48)))
print("... and over all blocks: mean %.3f; blocks where it is above 0.15: %s" % ((Gb >= 8).mean(), [int(b) for b in np.nonzero((Gb >= 8).mean(0) > 0.15)[0]]))
is_max = np.zeros(len(d), bool); is_max[np.arange(args.launches) * nblk + per.argmax(1)] = True
for gv in np.unique(G):
    m = G == gv
    print(f"   {int(gv):3d} groups: {int(m.sum()):6d} blocks  {d[m].mean():6.2f} us   {is_max[m].sum() / args.launches:5.2f}")
seg = np.stack([tk[:, 3], tk[:, 4] - tk[:, 3], tk[:, 6] - tk[:, 4], tk[:, 7] - tk[:, 6], tk[:, 0] - tk[:, 7]], 1)
if args.p_internal:
    slow = d >= np.percentile(d, 97); rest = d <= np.percentile(d, 60)
    names = ["entry state gathered + published", "action chosen + published", "own physics done, pair groups listed", "pooled pair groups done, results read back", "result line written",
             "trace, events, histogram issued", "phase-P barrier passed"]
    print(f"phase P of env wave 0 (s_memtime since entry), slowest 3 % of the blocks ({int(slow.sum())}) against the lower 60 % ({int(rest.sum())}): wall {d[slow].mean():.2f} / {d[rest].mean():.2f} us")
    prev_s = prev_r = 0.0
    for i, nm in enumerate(names, 1):
        a, b_ = tk[slow, i].mean(), tk[rest, i].mean()
        print(f"  {nm:44s} {a:9.0f} {b_:9.0f}   step {a - prev_s:7.0f} {b_ - prev_r:7.0f}  {(a - prev_s) - (b_ - prev_r):+7.0f}")
        prev_s, prev_r = a, b_
    print(f"  {'whole kernel':44s} {tk[slow, 0].mean():9.0f} {tk[rest, 0].mean():9.0f}")
    sys.exit(0)
lab = ["head (-> P barrier)", "Z + lists", "E (-> barrier)", "U2", "tail"]
slow = d >= np.percentile(d, 97); rest = d <= np.percentile(d, 60)
print(f"same launches, env wave 0's s_memtime ticks: slowest 3 % of the blocks ({int(slow.sum())}) against the lower 60 % ({int(rest.sum())})")
print(f"  wall time us                  {d[slow].mean():9.2f} {d[rest].mean():9.2f}   start offset within the launch us {t0[slow].mean():6.2f} {t0[rest].mean():6.2f}")
print(f"  whole kernel, ticks           {tk[slow, 0].mean():9.0f} {tk[rest, 0].mean():9.0f}   ticks per us {tk[slow, 0].mean() / d[slow].mean():7.1f} {tk[rest, 0].mean() / d[rest].mean():7.1f}")
for j, l in enumerate(lab):
    print(f"  {l:28s}  {seg[slow, j].mean():9.0f} {seg[rest, j].mean():9.0f}   {seg[slow, j].mean() - seg[rest, j].mean():+8.0f}")
bi = r[:, 6].astype(int)
cnt = np.bincount(bi[slow], minlength=nblk)
print("blocks most often among the slowest 3 %:", [(int(b), int(c)) for b, c in zip(np.argsort(cnt)[::-1][:14], np.sort(cnt)[::-1][:14])])
print("... by position in the env order (16 bins of 16 blocks), share of the slow set:", " ".join(f"{cnt[16 * k:16 * k + 16].sum() / max(cnt.sum(), 1):.2f}" for k in range(16)))
if args.rotate:
    pc = cnt[(np.arange(nblk) + args.rotate) % nblk]      # pc[p] = how often WORKGROUP p was slow
    print(f"(rotated by {args.rotate}: block = (workgroup + {args.rotate}) % {nblk}) by WORKGROUP position (16 bins):", " ".join(f"{pc[16 * k:16 * k + 16].sum() / max(pc.sum(), 1):.2f}" for k in range(16)))
    print("... by XCD of the WORKGROUP (workgroup % 8):", " ".join(f"{pc[x::8].sum() / max(pc.sum(), 1):.2f}" for x in range(8)))
print("... by XCD (block % 8):", " ".join(f"{cnt[x::8].sum() / max(cnt.sum(), 1):.2f}" for x in range(8)))

ex = np.stack(extra, 0).astype(np.float64)          # [launch, block, feature]
pb = per.mean(0)
print("per block, mean over the launches: wall us | entering an option | episodes begun | left an initiation set | came into one | env-id segments (of 256) the block draws from")
for b in list(range(20, 48)) + [100, 101, 102, 103]:
    print(f"  block {b:3d}  {pb[b]:6.2f} | " + " | ".join(f"{ex[:, b, j].mean():6.2f}" for j in range(ex.shape[2])))
odd = np.array([b for b in range(24, 42) if b % 2 == 1]); even = np.array([b for b in range(24, 42) if b % 2 == 0])
print(f"blocks 24..41: odd {pb[odd].mean():.2f} us, even {pb[even].mean():.2f} us; all blocks: odd {pb[1::2].mean():.2f}, even {pb[0::2].mean():.2f}")
