"""SkillChainingAgent — host driver of the fused step-batch (SPEC.md §4–§5).

north_star names `SkillChainingAgent.q_update` and asks that the Option / SkillChainingAgent Python API
be kept; the reference holds no code (README.md:1-2 only), so the API below is this build's own,
named after north_star. Per-step work is one scg_step launch pair (fused kernel + reduce/apply);
nothing on the per-step path synchronises with or copies to the host."""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import dist as _dist
from ._lib import CLF_STRIDE, NUM_ACTIONS, NUM_FEATURES, ScgError, auto_block_envs
from .core import EnvState, ScgContext
from .maps import PinballMap, load_map
from .option import Option
from .pinball import PinballDomain


class SkillChainingAgent:
    def __init__(self, pmap, n_envs: int, n_options: int = 0, *, device: int = 0, seed: int = 0,
                 env_id_base: int = 0, group=None, ordered_sum: bool = False, **hparams):
        self.map: PinballMap = load_map(pmap) if isinstance(pmap, str) else pmap
        if group is not None and hparams.get("block_envs") is None and not os.environ.get("SCG_BLOCK_ENVS"):
            # a sharded run uses ONE block geometry (it orders the partial sums of G): the ranks agree on the largest any of them
            # would pick from its own env count (equal shards pick equal sizes anyway)
            hparams["block_envs"] = _dist.allreduce_max_int(auto_block_envs(n_envs), group, torch.device("cuda", device))
        self.ctx = ScgContext(n_envs, n_options, self.map, device=device, seed=seed, env_id_base=env_id_base,
                              **hparams)
        dev = self.ctx.device
        self.n_envs, self.n_options, self.n_vf = n_envs, n_options, n_options + 1
        self.W = torch.zeros((self.n_vf, NUM_ACTIONS, NUM_FEATURES), dtype=torch.float32, device=dev)
        self.clf = torch.zeros((self.n_vf, CLF_STRIDE), dtype=torch.float32, device=dev)
        self.enabled_mask = 0
        self.gest_mask = 0            # SPEC §4.4: options in gestation
        self._gest_need = {}
        self.t = 0
        self.group = group            # torch.distributed group for shared option-Q weights (or None)
        self.ordered_sum = bool(ordered_sum)   # shared weights summed in rank order from an all-gather (identical on every
        self._slots = None                     # rank of the run, reproduced by the oracle) instead of an all-reduce (exact for two ranks)
        self.allreduce_timing = None  # see time_allreduce()
        self.domain = PinballDomain(self.ctx)
        self.state: EnvState = self.domain.state
        self.options: List[Option] = [Option(self, k) for k in range(self.n_vf)]

    # ------------------------------------------------------------------ option management (outer loop)
    def enable_option(self, k: int, enabled: bool = True) -> None:
        if not (1 <= k <= self.n_options):
            raise ValueError("option index out of range")
        self.enabled_mask = (self.enabled_mask | (1 << k)) if enabled else (self.enabled_mask & ~(1 << k))

    # ------------------------------------------------------------------ skill discovery (outer loop, SPEC §7)
    def enable_tracing(self, ring_len: int = 256, max_examples: int = 65536) -> None:
        """Attach the device-resident trajectory ring + per-step event flags (costs ~13 B/env-step) and the per-option
        example buffers (max_examples each) the device-side trigger appends to."""
        self.trace = self.ctx.set_trace_buffers(ring_len)
        self._ex_cap = int(max_examples)
        self._ex = {}                     # k -> (xy[cap, 2], label[cap], count[1], prev_in[N] or None), all on the device

    def _ex_buffers(self, k: int):
        if k not in self._ex:
            dev, cap = self.W.device, self._ex_cap
            parent = int(self.ctx.parents[k])
            self._ex[k] = (torch.zeros((cap, 2), dtype=torch.float32, device=dev),
                           torch.zeros(cap, dtype=torch.uint8, device=dev),
                           torch.zeros(1, dtype=torch.int32, device=dev),
                           torch.zeros(self.n_envs, dtype=torch.uint8, device=dev) if parent else None)
        return self._ex[k]

    def collect_examples(self, k: int, l_pos: int = 32, l_neg: int = 32) -> None:
        """Call after a step_batch while option k is being created: envs whose step ended inside option k's target
        region (goal disc if parent[k] = 0, else the parent's initiation set, on the step they ENTER it) append their
        last l_pos ring states as positives and the l_neg states before those as negatives to option k's example
        buffer. Selection, compaction and the gather run on the device (SPEC §7; one launch behind the step, whose
        commit rows leave the row totals of the announced trigger): nothing is read back, the host is not in the
        per-step path; examples_held(k) fetches the count when the outer loop wants it."""
        xy, lab, cnt, prev = self._ex_buffers(k)
        parent = int(self.ctx.parents[k])
        self.ctx.collect_examples(1 if parent == 0 else (1 << parent), prev, l_pos, l_neg, xy.view(-1), lab, cnt)

    def examples_held(self, k: int) -> int:
        """Examples in option k's buffer (one device->host read; node-wide total when the agent is sharded, so that
        every rank takes the same branch in the outer loop)."""
        got = int(self._ex_buffers(k)[2].item())
        if self.group is not None:
            got = _dist.allreduce_sum_int(got, self.group, self.W.device)
        return got

    def examples(self, k: int):
        """(xy[n, 2], label[n]) collected for option k so far (views of the device buffers)."""
        xy, lab, cnt, _ = self._ex_buffers(k)
        n = int(cnt.item())
        return xy[:n], lab[:n]

    def create_option(self, k: int, iters: int = 400, lr: float = 3.0, l2: float = 1e-4, gestation: int = 0) -> float:
        """Fit initiation classifier k on the collected examples (GPU logistic regression) and start its value function
        from the root's. With gestation = 0 the option is enabled at once; with gestation = G > 0 it first gestates
        (SPEC §4.4, Konidaris & Barto 2009): its classifier is in use, it is never selected, every transition from inside
        its initiation set updates its value function off-policy, and poll_gestation() enables it once G such
        transitions have reached its target. Returns the training accuracy."""
        self.ctx.disarm_collect()            # the collection for this option is over
        xy, lab = self.examples(k)
        if self.group is not None:           # fit on the examples of ALL ranks (rank order): identical classifiers everywhere
            xy, lab = _dist.allgather_rows(xy.contiguous(), self.group), _dist.allgather_rows(lab.contiguous(), self.group)
        clf = self.options[k].initiation_classifier
        err = None
        try:
            clf.fit(xy.contiguous(), lab.contiguous(), iters=iters, lr=lr, l2=l2)
        except ScgError as e:                # the fit gave up on the device (sticky status word): the row is untouched
            err = e
        if self.group is not None:           # a give-up is rank-local: agree on it before anyone acts on the classifier,
            bad = _dist.allreduce_sum_int(1 if err else 0, self.group, self.W.device)      # or the peers hang in the next collective
            if bad and err is None:
                raise ScgError(f"create_option({k}): the initiation-set fit gave up on {bad} other rank(s); no rank enables the option")
        if err is not None:
            raise err
        self.W[k].copy_(self.W[0])
        if gestation > 0:
            self._gest_need[k] = int(gestation)
            self.gest_mask |= 1 << k
            succ = self.ctx.set_gestation(self.gest_mask)
            succ[k] = 0
        else:
            self.enable_option(k)
        pred = clf.predict(xy[:, 0].contiguous(), xy[:, 1].contiguous())
        return float((pred == lab).float().mean())

    def poll_gestation(self) -> list:
        """Enable every gestating option whose success count has reached its requirement (one device->host read;
        counts are summed over the ranks of a sharded agent). Returns the options enabled by this call."""
        if not self.gest_mask:
            return []
        succ = self.ctx.set_gestation(self.gest_mask).cpu()
        done = []
        for k in range(1, self.n_options + 1):
            if not (self.gest_mask >> k) & 1:
                continue
            n = int(succ[k])
            if self.group is not None:
                n = _dist.allreduce_sum_int(n, self.group, self.W.device)
            if n >= self._gest_need[k]:
                self.gest_mask &= ~(1 << k)
                self.enable_option(k)
                done.append(k)
        if done:
            self.ctx.set_gestation(self.gest_mask)
        return done

    def chain_skills(self, steps_per_option: int = 300, min_examples: int = 2000, max_examples: int = 40000,
                     l_pos: int = 24, l_neg: int = 24, start_coverage: float = 0.5, poll_every: int = 8,
                     gestation: int = 0, gestation_steps: int = 200, **fit) -> list:
        """The outer loop of skill chaining (Konidaris & Barto 2009, the paper README.md:2 names), host-side
        policy over the device-resident pieces: for each not-yet-enabled option k in index order, run
        step-batches until enough trajectories have entered k's target (its parent in the skill graph),
        fit initiation set k on them, let it gestate (`gestation` successes, at most `gestation_steps` step-batches)
        or enable it at once, and stop once the start states are covered. The host looks at the device-side
        counters only every `poll_every` step-batches. Returns one report dict per created option."""
        report = []
        sx = torch.as_tensor(self.map.starts[:, 0].copy(), device=self.W.device)
        sy = torch.as_tensor(self.map.starts[:, 1].copy(), device=self.W.device)
        max_examples = min(max_examples, self._ex_cap)
        for k in range(1, self.n_options + 1):
            if ((self.enabled_mask | self.gest_mask) >> k) & 1:
                continue
            got = steps = 0
            while steps < steps_per_option and got < max_examples:
                for _ in range(min(poll_every, steps_per_option - steps)):
                    self.step_batch()
                    self.collect_examples(k, l_pos, l_neg)
                    steps += 1
                got = self.examples_held(k)
                self.poll_gestation()
            if got < min_examples:
                break
            acc = self.create_option(k, gestation=gestation, **fit)
            gsteps = 0
            while (self.gest_mask >> k) & 1 and gsteps < gestation_steps:
                for _ in range(poll_every):
                    self.step_batch()
                gsteps += poll_every
                self.poll_gestation()
            if (self.gest_mask >> k) & 1:            # did not see enough successes: enable anyway, as the paper's
                self.gest_mask &= ~(1 << k)          # fixed-length gestation period would
                self.ctx.set_gestation(self.gest_mask)
                self.enable_option(k)
            cov = float(self.options[k].initiation_classifier.predict(sx, sy).float().mean())
            report.append(dict(option=k, parent=int(self.ctx.parents[k]), steps=steps, examples=got,
                               accuracy=acc, start_coverage=cov, gestation_steps=gsteps))
            if cov >= start_coverage:
                break
        return report

    # ------------------------------------------------------------------ the skill graph (SPEC §4.2)
    def set_option_parents(self, parents) -> None:
        """parents[k] = the option whose initiation set option k chains to (0 = task goal); entry 0 ignored.
        The default is a chain; any acyclic assignment gives a skill tree rooted at the goal."""
        self.ctx.set_option_parents(parents)

    def skill_graph(self):
        """networkx.DiGraph of the discovered skills: node 0 = task goal, node k = option k (attrs: enabled,
        classifier weights), edge k -> parent[k] = "executing k leads into the initiation set of parent"."""
        import networkx as nx
        g = nx.DiGraph()
        g.add_node(0, kind="goal", target=tuple(self.map.target))
        clf = self.clf.cpu().numpy()
        for k in range(1, self.n_options + 1):
            g.add_node(k, kind="option", enabled=bool((self.enabled_mask >> k) & 1), classifier=clf[k, :6].tolist())
            g.add_edge(k, int(self.ctx.parents[k]))
        return g

    def init_weights(self, std: float = 1e-3, seed: int = 0) -> None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.W.copy_(torch.randn(self.W.shape, generator=g) * std)

    # ------------------------------------------------------------------ the hot path
    def step_batch(self, learn: bool = True) -> None:
        """One fused step-batch over all envs (act, physics, options, features, Q, TD, update)."""
        shared = self.group is not None and learn
        if shared:
            gp = self.ctx.grad_packed()                  # G and the update counts: ONE all-reduce operand
        self.ctx.step(self.state, self.W, self.clf, self.enabled_mask, self.t, learn=learn, apply=not shared)
        if shared:
            timing = self.allreduce_timing
            sample = timing is not None and (self.t % timing["every"]) == 0
            if sample:                                   # measurement hook (bench.py): events on the stream of use
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.ordered_sum:
                if self._slots is None:
                    import torch.distributed as dist
                    self._slots = torch.zeros((dist.get_world_size(self.group), gp.numel()), dtype=torch.float32, device=gp.device)
                _dist.allgather_packed(gp, self._slots, self.group)     # one all-gather; the sum is taken in rank order on every rank
            else:
                _dist.allreduce_packed(gp, self.group)      # RCCL over xGMI: one latency-bound 26 KB x n_vf message
            if sample:
                e1.record()
                timing["events"].append((e0, e1))
            if self.ordered_sum:
                self.ctx.apply_update_slots(self.W, self._slots)
            else:
                self.ctx.apply_update_packed(self.W, gp)
        self.t += 1

    def time_allreduce(self, every: int = 0) -> Optional[dict]:
        """every > 0: bracket every `every`-th shared-weights all-reduce with an event pair on the current stream
        (what the step's stream waits for: the collective as the step sees it, exposed). every = 0: stop and return
        {"samples", "mean_us", "max_us"} of what was recorded (None if nothing was)."""
        if every > 0:
            self.allreduce_timing = {"every": int(every), "events": []}
            return None
        timing, self.allreduce_timing = self.allreduce_timing, None
        if not timing or not timing["events"]:
            return None
        torch.cuda.synchronize(self.W.device)
        us = [a.elapsed_time(b) * 1e3 for a, b in timing["events"]]
        return {"samples": len(us), "mean_us": sum(us) / len(us), "max_us": max(us)}

    # ------------------------------------------------------------------ checkpoint / resume
    _STATE_FIELDS = ("x", "y", "vx", "vy", "option_id", "opt_steps", "ep_steps", "qcache", "action", "reward", "done")

    def state_dict(self) -> dict:
        """Everything a bit-identical continuation needs: the env SoA, W, the classifier table, the option graph, the
        enabled mask, the step counter (RNG streams are keyed by (seed, global env id, t): no RNG state to save) and,
        when tracing, the trajectory ring + events + the examples collected so far. Tensors are copied to the host."""
        d = {"format": 1, "n_envs": self.n_envs, "n_options": self.n_options, "t": int(self.t),
             "enabled_mask": int(self.enabled_mask), "parents": torch.as_tensor(self.ctx.parents.copy()),
             "W": self.W.cpu(), "clf": self.clf.cpu(),
             "state": {f: getattr(self.state, f).cpu() for f in self._STATE_FIELDS}}
        if getattr(self, "trace", None) is not None:
            ring_x, ring_y, events, ev_len = self.trace
            d["trace"] = {"ring_x": ring_x.cpu(), "ring_y": ring_y.cpu(), "events": events.cpu(), "ev_len": ev_len.cpu()}
            d["ex_cap"] = self._ex_cap
            d["examples"] = {int(k): (xy[: int(cnt.item())].cpu(), lab[: int(cnt.item())].cpu())
                             for k, (xy, lab, cnt, _) in self._ex.items()}
            d["prev_in"] = {int(k): v[3].cpu() for k, v in self._ex.items() if v[3] is not None}
        d["gest_mask"] = int(self.gest_mask)
        d["gest_need"] = {int(k): int(v) for k, v in self._gest_need.items()}
        if self.gest_mask:
            d["gest_succ"] = self.ctx.set_gestation(self.gest_mask).cpu()
        return d

    def load_state_dict(self, d: dict) -> None:
        if d.get("format") != 1 or d["n_envs"] != self.n_envs or d["n_options"] != self.n_options:
            raise ValueError("checkpoint does not match this agent (format / n_envs / n_options)")
        self.W.copy_(d["W"]); self.clf.copy_(d["clf"])
        for f in self._STATE_FIELDS:
            getattr(self.state, f).copy_(d["state"][f])
        self.t, self.enabled_mask = int(d["t"]), int(d["enabled_mask"])
        if self.n_options:
            self.ctx.set_option_parents([int(v) for v in d["parents"]])
        if "trace" in d:
            self.enable_tracing(int(d["trace"]["ring_x"].shape[0]), int(d["ex_cap"]))
            for name, buf in zip(("ring_x", "ring_y", "events", "ev_len"), self.trace):
                buf.copy_(d["trace"][name])
            for k, (xy, lab) in d["examples"].items():
                bxy, blab, cnt, prev = self._ex_buffers(int(k))
                bxy[: xy.shape[0]].copy_(xy); blab[: lab.shape[0]].copy_(lab); cnt.fill_(int(lab.shape[0]))
                if prev is not None and int(k) in d["prev_in"]:
                    prev.copy_(d["prev_in"][int(k)])
        self.gest_mask = int(d.get("gest_mask", 0))
        self._gest_need = {int(k): int(v) for k, v in d.get("gest_need", {}).items()}
        if self.gest_mask:
            self.ctx.set_gestation(self.gest_mask).copy_(d["gest_succ"])
        self.ctx.invalidate_order()      # option ids were written outside scg_step: the next step sorts afresh (same order)

    def save(self, path: str) -> None:
        torch.save(self.state_dict(), path)

    def load(self, path: str) -> None:
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    def q_update(self, k: int, s, action, r, cont, s_next, apply: bool = True) -> None:
        """Batched intra-option Q-learning update of VF k on explicit transitions (SPEC §5):
        delta = r + cont * max_a' Q_k(s',a') - Q_k(s,a);  W_k[a] += alpha/n * scale * sum delta*phi(s)."""
        self.ctx.q_update(k, s, action, r, cont, s_next, self.W, apply=apply)

    def rollout(self, steps: int, learn: bool = True) -> None:
        for _ in range(steps):
            self.step_batch(learn)
