"""ScgContext — thin, checked Python wrapper over the C-ABI (include/scg_abi.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation happens in the HIP
kernels behind libscg_hip.so. Operand shapes/dtypes/devices are validated on the host before any
launch (a kernel fault can take the whole GPU host down)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import (CLF_STRIDE, MAX_OPTIONS, NUM_ACTIONS, NUM_FEATURES, STEP_APPLY, STEP_LEARN, ScgConfig,
                   ScgError)
from .maps import PinballMap


def fourier_scale_table(order: int = 5, n_vars: int = 4) -> np.ndarray:
    """SPEC §3: scale_f = 1/||c||_2 (1 for c = 0), float64 then rounded; canonical feature order."""
    n = order + 1
    idx = np.arange(n ** n_vars)
    c = np.stack([(idx // n ** (n_vars - 1 - d)) % n for d in range(n_vars)], 1).astype(np.float64)
    norm = np.sqrt((c * c).sum(1))
    norm[0] = 1.0
    return (1.0 / norm).astype(np.float32)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class ScgContext:
    # r_option_success defaults to 0: with SPEC §4.2's value-gated entry an option is entered where its value function promises at
    # least the root's, and both estimate the TASK's return (an option that ends bootstraps from the root) — a completion bonus
    # inflates the option's side of that comparison (profiles/r05_oracle_chain_curves_*.txt: 10 000 collapses a seed, 0 is best)
    def __init__(self, n_envs: int, n_options: int, pmap: PinballMap, *, device: int = 0, seed: int = 0,
                 env_id_base: int = 0, gamma: float = 0.99, alpha: float = 1e-3, epsilon: float = 0.05,
                 r_option_success: float = 0.0, max_episode_steps: int = 10000,
                 max_option_steps: int = 250, update_count_floor: int = 0, reoffer_period: int = 4,
                 block_envs: Optional[int] = None,
                 library: Optional[str] = None):
        if not torch.cuda.is_available():
            raise ScgError("no GPU visible to torch: the HIP path cannot run and there is no CPU fallback")
        if not (0 <= n_options <= MAX_OPTIONS):
            raise ScgError(f"n_options must be in [0, {MAX_OPTIONS}]")
        if block_envs is None:                               # SPEC §5 geometry: SCG_BLOCK_ENVS pins it, else by the env count
            block_envs = int(os.environ["SCG_BLOCK_ENVS"]) if os.environ.get("SCG_BLOCK_ENVS") else _lib.auto_block_envs(n_envs)
        # one library per geometry (64 / 128 / 256 envs per block = per workgroup); `library`: a variant build of the same ABI
        self.lib = _lib.load(block_envs) if library is None else _lib.load(None, path=library)
        self.block_envs = int(self.lib.scg_block_envs())
        self.n_envs, self.n_options, self.n_vf = int(n_envs), int(n_options), int(n_options) + 1
        self.device = torch.device("cuda", device)
        self.map = pmap
        self.cfg = ScgConfig(n_envs=n_envs, n_options=n_options, fourier_order=_lib.FOURIER_ORDER,
                             device=device, env_id_base=env_id_base, seed=seed, gamma=gamma, alpha=alpha,
                             epsilon=epsilon, r_option_success=r_option_success,
                             max_episode_steps=max_episode_steps, max_option_steps=max_option_steps,
                             update_count_floor=update_count_floor, reoffer_period=reoffer_period)
        self._ctx = C.c_void_p()
        _lib.check(self.lib.scg_create(C.byref(self._ctx), C.byref(self.cfg)), None, "scg_create")
        self.scale = fourier_scale_table()
        self.parents = np.arange(-1, n_options, dtype=np.int32).clip(0)      # default chain k -> k-1
        edges, starts, sc = pmap.edges, np.ascontiguousarray(pmap.starts, np.float32), pmap.scalars
        self._call("scg_set_map", edges.ctypes.data_as(C.c_void_p), len(edges),
                   starts.ctypes.data_as(C.c_void_p), len(starts), sc.ctypes.data_as(C.c_void_p),
                   self.scale.ctypes.data_as(C.c_void_p))

    # ------------------------------------------------------------------ plumbing
    def _call(self, name: str, *args) -> None:
        _lib.check(getattr(self.lib, name)(self._ctx, *args), self._ctx, name)

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t: torch.Tensor, dtype: torch.dtype, numel: int, name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor) or t.dtype != dtype or t.device != self.device \
                or not t.is_contiguous() or t.numel() != numel:
            raise ScgError(f"{name}: expected contiguous {dtype} tensor with {numel} elements on {self.device}, "
                           f"got {getattr(t, 'dtype', type(t))} {tuple(getattr(t, 'shape', ()))} "
                           f"on {getattr(t, 'device', '?')}")
        return t

    def close(self) -> None:
        self._step_args = self._step_keep = self._armed = None
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.scg_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hparams(self, **kw) -> None:
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise ScgError(f"unknown hyper-parameter {k}")
            setattr(self.cfg, k, v)
        c = self.cfg
        self._call("scg_set_hparams", c.gamma, c.alpha, c.epsilon, c.r_option_success,
                   c.max_episode_steps, c.max_option_steps, c.update_count_floor, c.reoffer_period)

    # ------------------------------------------------------------------ fused step-batch
    def _step_flags(self, learn: bool, apply: bool) -> int:
        return (STEP_LEARN if learn else 0) | (STEP_APPLY if (learn and apply) else 0)

    def step(self, st: "EnvState", W: torch.Tensor, clf: torch.Tensor, enabled_mask: int, t: int,
             learn: bool = True, apply: bool = True) -> None:
        # the validated, pre-marshalled pointer arguments of the last call are reused while the same tensors come back
        # (one step is two kernel launches: the host side of a call matters in short runs)
        # (the key holds every tensor's storage address, not only the Python ids: `.data =` / `set_()` on the same object
        # swaps the storage under an unchanged id)
        key = (id(st), id(W), id(clf), st.x.data_ptr(), st.y.data_ptr(), st.vx.data_ptr(), st.vy.data_ptr(),
               st.option_id.data_ptr(), st.opt_steps.data_ptr(), st.ep_steps.data_ptr(), st.qcache.data_ptr(),
               st.action.data_ptr(), st.reward.data_ptr(), st.done.data_ptr(), W.data_ptr(), clf.data_ptr())
        cached = getattr(self, "_step_args", None)
        if cached is not None and cached[0] == key:
            flags = self._step_flags(learn, apply)
            _lib.check(self._step_fn(self._ctx, *cached[1], C.c_uint32(enabled_mask), C.c_uint64(t), C.c_uint32(flags),
                                     self._stream()), self._ctx, "scg_step")
            return
        N = self.n_envs
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        self._chk(st.x, f32, N, "x"); self._chk(st.y, f32, N, "y")
        self._chk(st.vx, f32, N, "vx"); self._chk(st.vy, f32, N, "vy")
        self._chk(st.option_id, i32, N, "option_id"); self._chk(st.opt_steps, i32, N, "opt_steps")
        self._chk(st.ep_steps, i32, N, "ep_steps"); self._chk(st.qcache, f32, NUM_ACTIONS * N, "qcache")
        self._chk(st.action, u8, N, "action"); self._chk(st.reward, f32, N, "reward")
        self._chk(st.done, u8, N, "done")
        self._chk(W, f32, self.n_vf * NUM_ACTIONS * NUM_FEATURES, "W")
        self._chk(clf, f32, self.n_vf * CLF_STRIDE, "clf")
        flags = self._step_flags(learn, apply)
        if st is not getattr(self, "_last_state", None):      # another state object (its memory may be recycled)
            self.invalidate_order()
            self._last_state = st
        ptrs = (_ptr(st.x), _ptr(st.y), _ptr(st.vx), _ptr(st.vy), _ptr(st.option_id), _ptr(st.opt_steps),
                _ptr(st.ep_steps), _ptr(st.qcache), _ptr(st.action), _ptr(st.reward), _ptr(st.done), _ptr(W), _ptr(clf))
        self._step_fn = self.lib.scg_step
        self._step_keep = (st, st.x, st.y, st.vx, st.vy, st.option_id, st.opt_steps, st.ep_steps, st.qcache, st.action,
                           st.reward, st.done, W, clf)    # keeps the ids (and the cached pointers) from being recycled
        self._step_args = (key, ptrs)
        self._call("scg_step", *ptrs, C.c_uint32(enabled_mask), C.c_uint64(t), C.c_uint32(flags), self._stream())

    def invalidate_order(self) -> None:
        """Tell the library that option ids were written outside scg_step (reset, restore): re-sort next step."""
        self._step_args = self._step_keep = None
        self._call("scg_invalidate_order")

    def set_option_parents(self, parents) -> None:
        """SPEC §4.2 option graph: parents[k] (k = 1..n_options) = option whose initiation set option k targets,
        0 = the task goal. Default is the chain k -> k-1. Validated (range, acyclic) by the library."""
        arr = np.zeros(self.n_options + 1, np.int32)
        arr[1:] = np.asarray(list(parents)[1:self.n_options + 1] if len(parents) > self.n_options
                             else list(parents), np.int32)[: self.n_options]
        self.parents = arr
        self._call("scg_set_option_parents", arr.ctypes.data_as(C.c_void_p))

    # ------------------------------------------------------------------ outer-loop support (SPEC §7)
    def set_trace_buffers(self, ring_len: int):
        """Allocate and attach the trajectory ring + event buffers; returns (ring_x, ring_y, events, ev_len).
        ring_len must be a power of two; ring_len = 0 detaches."""
        self._armed = None                 # the library drops an announced trigger with the buffers it refers to
        if ring_len == 0:
            self._call("scg_set_trace_buffers", None, None, 0, None, None)
            self._trace = None
            return None
        if ring_len < 1 or ring_len & (ring_len - 1):
            raise ScgError("ring_len must be a power of two")
        N = self.n_envs
        ring_x = torch.zeros((ring_len, N), dtype=torch.float32, device=self.device)
        ring_y = torch.zeros((ring_len, N), dtype=torch.float32, device=self.device)
        events = torch.zeros(N, dtype=torch.uint8, device=self.device)
        ev_len = torch.zeros(N, dtype=torch.int32, device=self.device)
        self._call("scg_set_trace_buffers", _ptr(ring_x), _ptr(ring_y), ring_len, _ptr(events), _ptr(ev_len))
        self._trace = (ring_x, ring_y, events, ev_len)
        return self._trace

    def harvest(self, sel_env: torch.Tensor, l_pos: int, l_neg: int):
        """Examples for the listed envs (int32, device) from the ring: (xy[n,L,2] f32, label[n,L] u8)."""
        if getattr(self, "_trace", None) is None:
            raise ScgError("harvest: trace buffers are not attached (set_trace_buffers)")
        ring_x, ring_y, _, ev_len = self._trace
        n = sel_env.numel()
        self._chk(sel_env, torch.int32, n, "sel_env")
        if n and (int(sel_env.min()) < 0 or int(sel_env.max()) >= self.n_envs):
            raise ScgError("harvest: env index out of range")
        L = l_pos + l_neg
        if l_pos < 0 or l_neg < 0 or L < 1:
            raise ScgError("harvest: need l_pos + l_neg >= 1")
        xy = torch.zeros((n, L, 2), dtype=torch.float32, device=self.device)
        lab = torch.zeros((n, L), dtype=torch.uint8, device=self.device)
        self._call("scg_harvest", n, _ptr(sel_env), _ptr(ring_x), _ptr(ring_y), ring_x.shape[0], _ptr(ev_len),
                   l_pos, l_neg, _ptr(xy), _ptr(lab), self._stream())
        return xy, lab

    def collect_examples(self, event_bits: int, prev_in: Optional[torch.Tensor], l_pos: int, l_neg: int,
                         ex_xy: torch.Tensor, ex_label: torch.Tensor, count: torch.Tensor, rearm: bool = True) -> None:
        """SPEC §7 device-side trigger: envs whose events byte has one of `event_bits` set (with prev_in: on the step the
        bit goes up) append their most recent ring states to ex_xy[cap, 2] / ex_label[cap] behind the count[0] examples
        already there (device int32[1], in/out). Two small launches (row totals, then prefix + gather) — one when the
        trigger was announced (`rearm`, scg_arm_collect: the step's own commit rows then leave the row totals); nothing
        comes back to the host. With `rearm` the library keeps the RAW device pointers of prev_in and count and reads
        them inside every later scg_step until disarm_collect() / set_trace_buffers(): this object holds references to
        both tensors for exactly that long, so dropping yours cannot leave the step reading freed memory."""
        if getattr(self, "_trace", None) is None:
            raise ScgError("collect_examples: trace buffers are not attached (set_trace_buffers)")
        cap = ex_label.numel()
        self._chk(ex_xy, torch.float32, 2 * cap, "ex_xy"); self._chk(ex_label, torch.uint8, cap, "ex_label")
        self._chk(count, torch.int32, 1, "count")
        if prev_in is not None:
            self._chk(prev_in, torch.uint8, self.n_envs, "prev_in")
        if l_pos < 0 or l_neg < 0 or l_pos + l_neg < 1 or not (0 < event_bits < 64):
            raise ScgError("collect_examples: bad argument")
        self._call("scg_collect_examples", C.c_uint32(event_bits), _ptr(prev_in), l_pos, l_neg, _ptr(ex_xy), _ptr(ex_label),
                   _ptr(count), cap, self._stream())
        if rearm:            # the same trigger will come again after the next step: let that step leave the row totals behind
            self._call("scg_arm_collect", C.c_uint32(event_bits), _ptr(prev_in), l_pos, l_neg, _ptr(count))
            self._armed = (prev_in, count)       # keeps the announced buffers alive while the library holds their addresses
        else:
            self.disarm_collect()

    def disarm_collect(self) -> None:
        self._call("scg_arm_collect", C.c_uint32(0), None, 0, 0, None)
        self._armed = None

    def set_gestation(self, gest_mask: int) -> torch.Tensor:
        """SPEC §4.4: options in gestation (known, never selected, learning off-policy). Returns the device int32[n_vf]
        success counters the fused step adds to (kept across calls; zero an entry when its option starts gestating)."""
        if not hasattr(self, "_gest_succ"):
            self._gest_succ = torch.zeros(self.n_vf, dtype=torch.int32, device=self.device)
        self._call("scg_set_gestation", C.c_uint32(gest_mask), _ptr(self._gest_succ))
        self.gest_mask = gest_mask
        return self._gest_succ

    def grad_buffers(self):
        """(G[n_vf,5,1296] float32, n_k[n_vf] int32): caller-owned torch tensors that scg_step(LEARN)
        fills with the rank-local gradient sum and update counts (the all-reduce operands, SPEC §5)."""
        if not hasattr(self, "_gbuf"):
            G = torch.zeros((self.n_vf, NUM_ACTIONS, NUM_FEATURES), dtype=torch.float32, device=self.device)
            n = torch.zeros((self.n_vf,), dtype=torch.int32, device=self.device)
            self._call("scg_set_grad_buffers", _ptr(G), _ptr(n))
            self._gbuf = (G, n)
            self.__dict__.pop("_gpacked", None)
        return self._gbuf

    def grad_packed(self) -> torch.Tensor:
        """One flat float32 tensor [n_vf*5*1296 + n_vf]: G followed by the update counts as floats — the single
        all-reduce operand of a sharded run with shared weights (scg_set_grad_buffer_packed)."""
        if not hasattr(self, "_gpacked"):
            gp = torch.zeros(self.n_vf * NUM_ACTIONS * NUM_FEATURES + self.n_vf, dtype=torch.float32, device=self.device)
            self._call("scg_set_grad_buffer_packed", _ptr(gp))
            self._gpacked = gp
            self.__dict__.pop("_gbuf", None)
        return self._gpacked

    def apply_update_packed(self, W: torch.Tensor, gp: torch.Tensor) -> None:
        self._chk(W, torch.float32, self.n_vf * NUM_ACTIONS * NUM_FEATURES, "W")
        self._chk(gp, torch.float32, self.n_vf * NUM_ACTIONS * NUM_FEATURES + self.n_vf, "G_packed")
        self._call("scg_apply_update_packed", _ptr(W), _ptr(gp), self._stream())

    def apply_update_slots(self, W: torch.Tensor, slots: torch.Tensor) -> None:
        """The order-pinned multi-rank update (SPEC §5): `slots` [n_ranks, n_vf*5*1296 + n_vf] holds every rank's packed operand
        (an all-gather of grad_packed()); G and the counts are summed in slot order: the weights are identical on every rank of a run and reproducible by the oracle for any number of ranks (the rank count, like the block size and the seed, is part of the run's identity)."""
        per = self.n_vf * NUM_ACTIONS * NUM_FEATURES + self.n_vf
        if slots.dim() != 2 or slots.shape[1] != per or not slots.is_contiguous():
            raise ScgError(f"slots must be a contiguous [n_ranks, {per}] tensor")
        self._chk(W, torch.float32, self.n_vf * NUM_ACTIONS * NUM_FEATURES, "W")
        self._chk(slots, torch.float32, slots.shape[0] * per, "slots")
        self._call("scg_apply_update_slots", _ptr(W), _ptr(slots), C.c_int32(slots.shape[0]), C.c_int64(per), self._stream())

    def apply_update(self, W: torch.Tensor, G: torch.Tensor, n_k: torch.Tensor) -> None:
        self._chk(W, torch.float32, self.n_vf * NUM_ACTIONS * NUM_FEATURES, "W")
        self._chk(G, torch.float32, self.n_vf * NUM_ACTIONS * NUM_FEATURES, "G")
        self._chk(n_k, torch.int32, self.n_vf, "n_k")
        self._call("scg_apply_update", _ptr(W), _ptr(G), _ptr(n_k), self._stream())

    # ------------------------------------------------------------------ un-fused entry points
    def _chk4(self, s: Sequence[torch.Tensor], n: int, name: str):
        if len(s) != 4:
            raise ScgError(f"{name}: need (x, y, vx, vy)")
        return [self._chk(t, torch.float32, n, f"{name}[{i}]") for i, t in enumerate(s)]

    def pinball_step(self, s, action: torch.Tensor):
        n = s[0].numel()
        x, y, vx, vy = self._chk4(s, n, "state")
        self._chk(action, torch.uint8, n, "action")
        if n and int(action.max()) >= NUM_ACTIONS:
            raise ScgError("action out of range [0, 5)")
        reward = torch.empty(n, dtype=torch.float32, device=self.device)
        goal = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._call("scg_pinball_step", n, _ptr(x), _ptr(y), _ptr(vx), _ptr(vy), _ptr(action), _ptr(reward),
                   _ptr(goal), self._stream())
        return reward, goal

    def features(self, s) -> torch.Tensor:
        n = s[0].numel()
        x, y, vx, vy = self._chk4(s, n, "state")
        phi = torch.empty((n, NUM_FEATURES), dtype=torch.float32, device=self.device)
        self._call("scg_fourier_features", n, _ptr(x), _ptr(y), _ptr(vx), _ptr(vy), _ptr(phi), self._stream())
        return phi

    def q_values(self, s, Wk: torch.Tensor) -> torch.Tensor:
        n = s[0].numel()
        x, y, vx, vy = self._chk4(s, n, "state")
        self._chk(Wk, torch.float32, NUM_ACTIONS * NUM_FEATURES, "Wk")
        q = torch.empty((NUM_ACTIONS, n), dtype=torch.float32, device=self.device)
        self._call("scg_q_values", n, _ptr(x), _ptr(y), _ptr(vx), _ptr(vy), _ptr(Wk), _ptr(q), self._stream())
        return q

    def q_update(self, k: int, s, action, r, cont, sn, W: torch.Tensor, apply: bool = True) -> None:
        n = s[0].numel()
        if n > self.n_envs:
            raise ScgError("q_update: more transitions than the context's n_envs")
        x, y, vx, vy = self._chk4(s, n, "s")
        xn, yn, vxn, vyn = self._chk4(sn, n, "s_next")
        self._chk(action, torch.uint8, n, "action"); self._chk(r, torch.float32, n, "r")
        self._chk(cont, torch.float32, n, "cont")
        self._chk(W, torch.float32, self.n_vf * NUM_ACTIONS * NUM_FEATURES, "W")
        if n and int(action.max()) >= NUM_ACTIONS:
            raise ScgError("action out of range [0, 5)")
        self._call("scg_q_update", n, k, _ptr(x), _ptr(y), _ptr(vx), _ptr(vy), _ptr(action), _ptr(r), _ptr(cont),
                   _ptr(xn), _ptr(yn), _ptr(vxn), _ptr(vyn), _ptr(W), C.c_uint32(STEP_APPLY if apply else 0),
                   self._stream())

    def classifier_predict(self, x: torch.Tensor, y: torch.Tensor, w8: torch.Tensor) -> torch.Tensor:
        n = x.numel()
        self._chk(x, torch.float32, n, "x"); self._chk(y, torch.float32, n, "y")
        self._chk(w8, torch.float32, CLF_STRIDE, "w8")
        out = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._call("scg_classifier_predict", n, _ptr(x), _ptr(y), _ptr(w8), _ptr(out), self._stream())
        return out

    def fit_initiation(self, xy: torch.Tensor, label: torch.Tensor, offsets: torch.Tensor, w: torch.Tensor,
                       iters: int = 200, lr: float = 1.0, l2: float = 1e-4) -> None:
        n_fit = offsets.numel() - 1
        self._chk(offsets, torch.int32, n_fit + 1, "offsets")
        off = offsets.cpu()
        m = int(off[-1])
        if n_fit < 0 or int(off[0]) != 0 or bool((off[1:] < off[:-1]).any()):
            raise ScgError("offsets must start at 0 and be non-decreasing")
        self._chk(xy, torch.float32, 2 * m, "xy"); self._chk(label, torch.uint8, m, "label")
        self._chk(w, torch.float32, n_fit * CLF_STRIDE, "w")
        self._call("scg_fit_initiation", n_fit, _ptr(xy), _ptr(label), _ptr(offsets), _ptr(w), iters,
                   C.c_float(lr), C.c_float(l2), self._stream())
        # a fit whose workgroups could not run together gives up on the device (rows of w untouched) and says so in the
        # ctx's status word: the outer loop is about to act on these classifiers, so wait and look now (rare call)
        self.async_status(synchronize=True)

    # ------------------------------------------------------------------ asynchronous failures (include/scg_abi.h)
    def async_status(self, synchronize: bool = False) -> int:
        """Raise ScgError if a kernel launched earlier gave up on the device (sticky until clear_async_error);
        returns the raw status word otherwise (0)."""
        word = C.c_uint32(0)
        self._call("scg_async_status", self._stream(), 1 if synchronize else 0, C.byref(word))
        return int(word.value)

    def clear_async_error(self) -> None:
        self._call("scg_clear_async_error")

    def set_fit_timeout(self, seconds: float) -> None:
        """How long fit_initiation waits for a workgroup that is not running yet before it abandons that fit."""
        self._call("scg_set_fit_timeout", C.c_double(seconds))


class EnvState:
    """SoA env batch in HBM (caller-owned torch tensors)."""

    def __init__(self, n: int, device: torch.device, pmap: PinballMap):
        z = lambda dt: torch.zeros(n, dtype=dt, device=device)
        sx, sy = float(pmap.starts[0][0]), float(pmap.starts[0][1])
        self.n = n
        self.x = torch.full((n,), sx, dtype=torch.float32, device=device)
        self.y = torch.full((n,), sy, dtype=torch.float32, device=device)
        self.vx, self.vy = z(torch.float32), z(torch.float32)
        self.option_id, self.opt_steps, self.ep_steps = z(torch.int32), z(torch.int32), z(torch.int32)
        self.qcache = torch.zeros((NUM_ACTIONS, n), dtype=torch.float32, device=device)
        self.action, self.done = z(torch.uint8), z(torch.uint8)
        self.reward = z(torch.float32)

    def state(self):
        return (self.x, self.y, self.vx, self.vy)
