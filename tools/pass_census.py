import os, sys, numpy as np, torch
sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
n=65536; nopt=5
ag=SkillChainingAgent(bench.MAP,n,nopt,seed=0,**bench.HP)
ag.clf.copy_(torch.as_tensor(bench.chain_discs(ag.map,nopt)))
for k in range(1,nopt+1): ag.enable_option(k)
ag.init_weights(std=1e-3); ag.domain.reset_random(seed=1000)
for _ in range(200): ag.step_batch()
o=ag.state.option_id.cpu().numpy().copy()
ag.step_batch()
on=ag.state.option_id.cpu().numpy().copy()
B=ag.ctx.lib.scg_block_envs()
tot=np.bincount(o,minlength=7); S=tot[1:].sum(); R=(tot[1:]>0).sum(); Bf=n//B
c=min(B,-(-S//(Bf-R)))
lst=[np.nonzero(o==k)[0] for k in range(7)]
perm=[]; fill=0
for k in range(1,7):
    for r in range(0,tot[k],c):
        ch=lst[k][r:r+c]; perm+=list(ch); m=B-len(ch); perm+=list(lst[0][fill:fill+m]); fill+=m
perm+=list(lst[0][fill:]); perm=np.array(perm)
assert len(perm)==n and len(set(perm))==n
nb=n//B; upd_passes=0; eval_only=0; hist=np.zeros(8,int); items=[]
for b in range(nb):
    e=perm[b*B:(b+1)*B]
    up=set(o[e].tolist())|{0}
    ev=set(on[e].tolist())
    eo=len(ev-up); eval_only+=eo; upd_passes+=len(up); hist[eo]+=1
    for v in ev-up: items.append(int((on[e]==v).sum()))
print("blocks",nb,"chunk c",c,"update passes/block",upd_passes/nb,"eval-only passes/block",eval_only/nb,"hist of eval-only per block",hist.tolist())
print("option counts",tot.tolist())

items=np.array(items); print("items per evaluation-only pass: mean %.1f, median %d, p90 %d, max %d; quads mean %.2f" % (items.mean(), np.median(items), np.percentile(items,90), items.max(), np.ceil(items/4).mean()))
