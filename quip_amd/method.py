"""`QuantMethod` and the structured random-orthogonal operators -- the surface of the reference's method.py
(method.py:16-233).  Same class / function names, call order and side effects (opt.py:97-170 drives it as
add_batch* -> post_batch -> preproc -> fasterquant -> free); what changes is how the work is done:

  * the incoherence projection is applied in its factored (butterfly / Kronecker) form by
    quip_amd/csrc/ortho.hip -- the dense n x n matrices U, V of method.py:162-176 are never built;
  * `projU` / `projV` therefore hold the generator tuples (B, p_in, p_out), not dense matrices.

Random state is consumed exactly like the reference (U then V; factors through
scipy.stats.special_ortho_group / torch.rand, then two torch.randperm), so seeding numpy + torch the same way
gives the same operators.
"""
import math

import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import transformers

from . import ops

DEBUG = False
FUSED_PREPROC = True      # False: preproc's rescale / trace-ridge chains as the torch elementwise ops they are in the reference
HESSIAN_FAST = False     # True: K7's opt-in 16-bit-MFMA mode for add_batch (f16 / bf16 inputs; see include/quip_amd.h)
DEVICE_RNG = False       # True: draw the Gaussians of the SO(p) sampler with torch's device generator instead of numpy's
                         # legacy stream -- same distribution (Haar), NOT the reference's seeded operators; removes the
                         # host RNG that bounds preproc (0.8 M / 3.8 M normals per n = 8192 / 11008 operator)


SHARE_IDENTICAL_INPUTS = True   # Linears that are handed THE SAME input tensor (q / k / v of an attention block, gate / up of a gated MLP) accumulate
                                # its X^T X once: the later ones count samples and take a copy of the first one's Hessian at post_batch
                                # (bit-identical to accumulating it again -- same numbers, same kernel, same order -- at a third of the work)
OPERATOR_PREFETCH = False  # True: sample the operators of the Linears whose QuantMethod objects already exist on a host thread (below).
                           # Exact -- the same draws from the same streams -- PROVIDED nobody reseeds numpy / torch between constructing a method
                           # and its preproc (a draw made ahead of a reseed would be taken as if made behind it).  The reference's drivers
                           # never do; tests and notebooks do.  So the library default is off and the whole-model drivers of this repo
                           # (scripts/run_full_model.py, bench.py's quantise_model leg) switch it on.
                           # Second restriction: `unbiased=True` draws torch.rand on the CPU global generator INSIDE fasterquant
                           # (vector_balance.py:174-175), i.e. between two preprocs -- a prefetched draw would then sit on the wrong side of
                           # it.  vector_balance._draw_eta therefore DRAINS the prefetcher (rewinds both generators to where a run without it would stand and
                           # forgets the queue) before it draws; the Linears queued behind it sample synchronously.


def _prime_factors(n):
    """ascending prime factors (stands in for primefac.primefac, method.py:17; not installed here)."""
    out, f = [], 2
    while f * f <= n:
        while n % f == 0:
            out.append(f)
            n //= f
        f += 1 if f == 2 else 2
    if n > 1:
        out.append(n)
    return out


def butterfly_factors(n):
    """n = p*q with p the product of the even-indexed prime factors, q of the odd-indexed (method.py:16-18)."""
    pf = _prime_factors(n)
    return (math.prod(pf[0::2]), math.prod(pf[1::2]))


def _special_ortho_group_gpu(p, m, dev):
    """scipy.stats.special_ortho_group.rvs(p, size=m) with the Householder accumulation on the GPU.

    scipy (1.15, unpinned by the reference) builds each sample as H = D * prod_n (I - x_n x_n^T): for n = 0 .. p-2 it
    draws x_n ~ N(0, I_{p-n}) from the global numpy RandomState, turns it into the Householder vector of the reflection
    sending x_n to -sign(x_n[0]) |x_n| e_0, applies it to columns n: of H, and finally scales the rows by D (signs, last
    one chosen so that det = +1).  Its Python loop over n with numpy broadcasting takes 0.5 s for 64 x SO(128) -- most
    of QuantMethod.preproc's time once the projection itself runs on K3.  Here the SAME normals are drawn from the
    SAME stream in the same order (so seeding numpy reproduces the reference's operators), the vector preparation is
    done for all n at once, and the p-1 rank-one updates are applied 64 at a time in compact-WY form.
    Differences to scipy are fp64 summation order only (~1e-16), invisible after the fp32 narrowing of method.py:22."""
    shape = (m,) if m > 1 else ()
    if DEVICE_RNG and dev.type == 'cuda':
        x = torch.randn(p - 1, max(m, 1), p, dtype=torch.float64, device=dev)
        x = x * (torch.arange(p, device=dev)[None, None, :] >= torch.arange(p - 1, device=dev)[:, None, None])
    else:
        xs = np.zeros((p - 1, max(m, 1), p))
        for n in range(p - 1):
            xs[n, :, n:] = np.random.normal(size=shape + (p - n,)).reshape(max(m, 1), p - n)    # same calls as scipy
        x = torch.from_numpy(xs).to(dev)                                # [p-1, m, p], x_n zero-padded below column n
    norm2 = (x * x).sum(-1)
    idx = torch.arange(p - 1, device=dev)
    x0 = x[idx, :, idx].clone()                                         # [p-1, m]: leading element of every x_n
    D = torch.where(x0 != 0, torch.sign(x0), torch.ones_like(x0))
    lead = x0 + D * norm2.sqrt()
    x[idx, :, idx] = lead
    x = x / ((norm2 - x0 * x0 + lead * lead) / 2.).sqrt()[..., None]
    # H = prod_n (I - x_n x_n^T), applied from the right in order.  A block of b reflectors is I - X T X^T with
    # T^-1 = I + strict_upper(X^T X) (compact WY for unit tau), so each block costs five batched launches instead of 2 b:
    # p = 688 (Llama MLP width) went from 1374 launches to 55.  Same product up to fp64 round-off.
    H = torch.eye(p, dtype=torch.float64, device=dev).repeat(max(m, 1), 1, 1)
    blk = 64
    eye_b = torch.eye(blk, dtype=torch.float64, device=dev)
    for s0 in range(0, p - 1, blk):
        X = x[s0:min(s0 + blk, p - 1)].permute(1, 2, 0)                  # [m, p, b]
        b = X.shape[2]
        G = torch.bmm(X.transpose(1, 2), X)                              # [m, b, b]
        Tinv = torch.triu(G, diagonal=1) + eye_b[:b, :b]
        HX = torch.bmm(H, X)                                             # [m, p, b]
        HXT = torch.linalg.solve_triangular(Tinv, HX, upper=True, left=False)   # (H X) T
        H = torch.baddbmm(H, HXT, X.transpose(1, 2), alpha=-1.0)
    Dlast = (-1) ** (p - 1) * D.prod(0)
    H = H * torch.cat([D, Dlast[None]], 0).t()[:, :, None]
    return H if m > 1 else H[0]


def gen_rand_orthos(m, p):
    """m Haar-random SO(p) matrices as fp32 (method.py:20-31).  p == 2 draws rotation angles from torch.rand.
    The random stream is consumed exactly like the reference; with a GPU present the accumulation of scipy's
    Householder factors runs there (_special_ortho_group_gpu), otherwise scipy itself is called."""
    if p != 2:
        if torch.cuda.is_available() and p > 2:
            return _special_ortho_group_gpu(p, m, torch.device('cuda', torch.cuda.current_device())).to(torch.float32).cpu()
        return torch.tensor(scipy.stats.special_ortho_group.rvs(p, size=m)).to(torch.float32)
    theta = torch.rand(m) * (2 * math.pi)
    c, s = torch.cos(theta), torch.sin(theta)
    return torch.stack([torch.stack([c, s], -1), torch.stack([-s, c], -1)], -2)


def gen_rand_ortho_butterfly(n):
    """blocked two-factor butterfly + permutations (method.py:34-35)."""
    return ([gen_rand_orthos(n // p, p) for p in butterfly_factors(n)], torch.randperm(n), torch.randperm(n))


def gen_rand_ortho_butterfly_noblock(n):
    """one factor per stage = Kronecker product (method.py:38-39)."""
    return ([gen_rand_orthos(1, p) for p in butterfly_factors(n)], torch.randperm(n), torch.randperm(n))


def gen_rand_ortho_butterfly_nopermute(n):
    """blocked, identity permutations (method.py:42-43)."""
    return ([gen_rand_orthos(n // p, p) for p in butterfly_factors(n)], torch.arange(n), torch.arange(n))


def mul_ortho_butterfly(Bpp, x):
    """Q @ x for x [n] or [n, c] (method.py:46-67), on the GPU through K3."""
    one_d = x.dim() == 1
    xc = x.reshape(x.shape[0], -1)
    dev = xc.device if xc.is_cuda else torch.device('cuda:0')
    y = ops.OrthoOp(Bpp, dev).apply_cols(xc.to(dev, torch.float32)).to(x.device)
    return y.reshape(-1) if one_d else y


def rand_ortho_butterfly(n):
    """dense matrix of a fresh operator (method.py:71-72); kept for API parity -- preproc does not use it."""
    return mul_ortho_butterfly(gen_rand_ortho_butterfly(n), torch.eye(n))


def rand_ortho_butterfly_noblock(n):
    return mul_ortho_butterfly(gen_rand_ortho_butterfly_noblock(n), torch.eye(n))


def rand_ortho_butterfly_nopermute(n):
    return mul_ortho_butterfly(gen_rand_ortho_butterfly_nopermute(n), torch.eye(n))


_GENERATORS = {0: gen_rand_ortho_butterfly, 1: gen_rand_ortho_butterfly_noblock, 2: gen_rand_ortho_butterfly_nopermute}


class _OperatorPrefetcher:
    """Operator sampling off the critical path, on the SAME random streams (opt-in: OPERATOR_PREFETCH).

    `gen_rand_orthos` is host work on numpy's legacy Gaussian stream (kept for seed parity with the reference, method.py:20-31) plus a
    chain of small batched GEMMs: 80 ms per OPT-1.3B block, 0.23 s per Llama-2-7B block (profiles/r04a: 7.3 of 58 s) -- all of it inside
    `preproc`, between two GPU phases, with the GPU idle.  The reference's drivers create every QuantMethod of a block BEFORE they run the
    block's calibration forwards (opt.py:99-129 then :141-143; llama.py:91-116 then :130-131) and call `preproc` on them in that same order
    (opt.py:151, llama.py:138).  So when a QuantMethod is constructed its (rows, columns) are queued here, and a host thread draws the
    operators of the queued Linears -- in queue order, U before V, exactly the draws `preproc` would make (numpy's and torch's global
    generators are only ever touched by these draws while a driver runs) -- while the main thread feeds the GPU with the block's forwards
    and Hessian launches.  `preproc` then takes its pair from the head of the queue.

    What the thread cannot know is the generator `preproc` will be asked for (`preproc_proj_extra`) or whether it will project at all: it
    assumes the arguments of the previous `preproc` call (nothing is prefetched before the first one).  Every speculative draw is preceded
    by a snapshot of both generator states; a `preproc` that does not find what it needs at the head of the queue (other flags, another
    call order, a method that was dropped) rewinds numpy and torch to the snapshot in front of the oldest unconsumed draw, forgets the
    queue and samples synchronously -- the streams are then exactly where the reference's would be.  Seeded runs therefore reproduce the
    same operators with and without the prefetcher (tests/test_host_logic.py)."""

    def __init__(self):
        import threading
        self.lock = threading.Condition()
        self.pending = []            # [(key, rows, cols)] not yet sampled
        self.ready = []              # [(key, extra, projU, projV, snapshot)] sampled, in draw order
        self.busy = None             # key being sampled
        self.flags = None            # (extra,) of the last projecting preproc call, or None
        self.thread = None
        self.stop = False
        self.stats = {"prefetched": 0, "rewinds": 0, "sync": 0, "wait_s": 0.0}

    # -- main thread ---------------------------------------------------------------------------------------------------------------
    def request(self, key, rows, cols):
        import threading
        with self.lock:
            if self.flags is None:
                return
            self.pending.append((key, rows, cols))
            if self.thread is None or not self.thread.is_alive():
                self.stop = False
                # a new thread's current device is cuda:0 whatever the creating thread selected: hand ours over (rank r of an N-GPU run
                # must sample on GPU r, not pile its side stream onto GPU 0)
                self.device = torch.cuda.current_device() if torch.cuda.is_available() else None
                self.thread = threading.Thread(target=self._run, name="quip_amd-operator-prefetch", daemon=True)
                self.thread.start()
            self.lock.notify_all()

    def take(self, key, extra):
        """(projU, projV) for the method `key` drawn with generator `extra`, or None after rewinding (the caller samples itself)"""
        import time
        t0 = time.perf_counter()
        with self.lock:
            while True:
                if self.ready and self.ready[0][0] == key and self.ready[0][1] == extra:
                    _, _, U, V, _ = self.ready.pop(0)
                    self.stats["prefetched"] += 1
                    self.stats["wait_s"] += time.perf_counter() - t0
                    return U, V
                head_is_mine = (not self.ready and ((self.busy == key) or (self.busy is None and self.pending and self.pending[0][0] == key)))
                if head_is_mine and self.flags == (extra,):
                    self.lock.wait(0.05)                 # being sampled / next in line with the right generator: wait for it
                    continue
                self._rewind_locked()
                self.stats["sync"] += 1
                return None

    def note_flags(self, proj, extra):
        with self.lock:
            self.flags = (extra,) if proj else None

    def drain(self):
        """forget everything queued and rewind the generators to where a run without the prefetcher would have left them"""
        with self.lock:
            self._rewind_locked()

    def shutdown(self):
        with self.lock:
            self._rewind_locked()
            self.stop = True
            self.lock.notify_all()
        if self.thread is not None:
            self.thread.join(timeout=10)

    def _rewind_locked(self):
        self.pending.clear()
        while self.busy is not None:                     # let the draw in flight finish: its snapshot is the one to go back to
            self.lock.wait(0.05)
        if self.ready:
            np_state, torch_state = self.ready[0][4]
            np.random.set_state(np_state)
            torch.set_rng_state(torch_state)
            self.ready.clear()
            self.stats["rewinds"] += 1

    # -- the thread ------------------------------------------------------------------------------------------------------------------
    def _run(self):
        if getattr(self, "device", None) is not None:
            torch.cuda.set_device(self.device)
        side = torch.cuda.Stream() if torch.cuda.is_available() else None
        while True:
            with self.lock:
                while not self.pending and not self.stop:
                    self.lock.wait(0.5)
                if self.stop:
                    return
                key, rows, cols = self.pending.pop(0)
                extra = self.flags[0] if self.flags is not None else None
                if extra is None:
                    continue
                self.busy = key
                snap = (np.random.get_state(), torch.get_rng_state())
            try:
                gen = _GENERATORS[extra]
                if side is not None:
                    with torch.cuda.stream(side):
                        U, V = gen(rows), gen(cols)
                else:
                    U, V = gen(rows), gen(cols)
                item = (key, extra, U, V, snap)
            except Exception:                              # a failed draw: hand the stream position back, preproc samples itself
                item = None
                np.random.set_state(snap[0])
                torch.set_rng_state(snap[1])
            with self.lock:
                if item is not None:
                    self.ready.append(item)
                self.busy = None
                self.lock.notify_all()


_prefetcher = None


def operator_prefetcher():
    global _prefetcher
    if _prefetcher is None:
        _prefetcher = _OperatorPrefetcher()
    return _prefetcher


def operator_prefetch_stop():
    """stop the prefetch thread and rewind the generators past nothing that was not consumed (call when a driver run ends)"""
    global _prefetcher
    if _prefetcher is not None:
        _prefetcher.shutdown()
        _prefetcher = None


_last_inputs = {}      # input signature -> (weakref to the method that accumulated it last, the tensor itself, that method's nsamples afterwards)


def _input_signature(t):
    return (str(t.device), int(t.data_ptr()), tuple(t.shape), tuple(t.stride()), t.dtype)


class QuantMethod:
    """Base class for the rounding methods (method.py:80-233)."""

    def __init__(self, layer):
        self.layer = layer
        self.dev = self.layer.weight.device
        W = layer.weight.data
        if isinstance(self.layer, nn.Conv2d):
            W = W.flatten(1)
        if isinstance(self.layer, transformers.Conv1D):
            W = W.t()
        self.rows, self.columns = W.shape[0], W.shape[1]
        self.H = torch.zeros((self.columns, self.columns), dtype=torch.float64, device=self.dev)
        self.nsamples = 0
        self.preproc_done = False
        if OPERATOR_PREFETCH and not isinstance(layer, nn.Conv2d):
            operator_prefetcher().request(id(self), self.rows, self.columns)

    # ---- Hessian accumulation (method.py:98-123) ------------------------------------------------------
    # On the GPU, Linear / Conv1D inputs go through K7 (quip_amd/csrc/hessian.hip): fp64 X^T X of the block-lower
    # triangle on the fp64 matrix pipe straight from the token-major hook input (no transpose, no fp64 copy of X);
    # `H` then holds the triangle only until post_batch mirrors it.  `_tri` records that state; assigning `.H` from
    # outside (optq_ldlq_equiv.py:24,35) leaves it False, and post_batch then takes the reference's dense route.
    def share_hessian_from(self, leader):
        """This layer sees the SAME input as `leader` (q/k/v of an attention block, gate/up of a gated MLP): the
        reference accumulates the identical X^T X once per Linear (opt.py:131-145); after this call
        add_batch here only counts samples and post_batch takes its own fp32 copy of the leader's Hessian (preproc
        then modifies it per layer, method.py:139-176).  Call before the first add_batch."""
        assert leader is not self and self.nsamples == 0 and leader.columns == self.columns
        self._leader = leader
        leader._followers = getattr(leader, "_followers", 0) + 1
        self.H = None

    def add_batch(self, inp, out):
        if DEBUG:
            self.inp1, self.out1 = inp, out
        keep = inp                                 # (the tensor as it was handed in: what a sibling Linear would be handed too)
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        n_calls = inp.shape[0]                     # nsamples counts hook calls' batch dim, not tokens
        if getattr(self, "_leader", None) is not None:
            self.nsamples += n_calls
            return
        linear = isinstance(self.layer, (nn.Linear, transformers.Conv1D))
        sig = None
        if SHARE_IDENTICAL_INPUTS and type(self.layer) is nn.Linear and inp.dtype in (torch.float16, torch.bfloat16, torch.float32):
            # The drivers hook every Linear of a block (opt.py:131-140) and HF hands q / k / v (gate / up) the very same tensor: its X^T X is
            # identical, bit for bit, for all of them.  The method that sees a tensor FIRST accumulates it and keeps it alive until its own
            # next sample (so the address cannot be handed to another tensor); a method whose every call so far found its input already
            # accumulated by that same leader, sample for sample, never accumulates.  (Assumes nobody rewrites the shared input in place
            # between the sibling Linears -- HF's OPT / Llama blocks do not; SHARE_IDENTICAL_INPUTS = False restores one X^T X per Linear.)
            sig = _input_signature(keep)
            ent = _last_inputs.get(sig)
            lead = ent[0]() if ent is not None else None
            auto = getattr(self, "_auto_leader", None)
            if auto is not None:
                if lead is auto and ent[2] == self.nsamples + n_calls:
                    self.nsamples += n_calls
                    return
                raise RuntimeError("add_batch: this layer shared its input with another one so far (method.SHARE_IDENTICAL_INPUTS) and now "
                                   "receives a different tensor; set quip_amd.method.SHARE_IDENTICAL_INPUTS = False for this model")
            if (self.nsamples == 0 and getattr(self, "_followers", 0) == 0 and lead is not None and lead is not self
                    and lead.columns == self.columns and ent[2] == n_calls
                    and getattr(lead, "_auto_leader", None) is None and getattr(lead, "_leader", None) is None
                    and lead.H is not None and lead.H.dtype == torch.float64 and type(lead.layer) is nn.Linear):
                self._auto_leader = lead
                lead._followers = getattr(lead, "_followers", 0) + 1
                self.H = None
                self.nsamples += n_calls
                return
        if linear and inp.dim() == 3:
            inp = inp.reshape(-1, inp.shape[-1])
        if (linear and inp.is_cuda and inp.dim() == 2 and inp.dtype in ops._DT and self.H.dtype == torch.float64
                and (self.nsamples == 0 or getattr(self, "_tri", False))):
            self.nsamples += n_calls
            self._tri = True
            ops.hessian_accum(self.H, inp, fast=HESSIAN_FAST)   # raises if the HIP library is missing: no fallback on the GPU
            self._remember_input(sig, keep)
            return
        if getattr(self, "_tri", False):
            raise RuntimeError("add_batch: Hessian accumulation started on the HIP path; cannot mix input kinds")
        if linear:
            inp = inp.t()
        elif isinstance(self.layer, nn.Conv2d):
            unfold = nn.Unfold(self.layer.kernel_size, dilation=self.layer.dilation, padding=self.layer.padding,
                               stride=self.layer.stride)
            inp = unfold(inp).permute(1, 0, 2).flatten(1)
        self.nsamples += n_calls
        inp = inp.to(torch.float64)
        self.H.addmm_(inp, inp.t())
        self._remember_input(sig, keep)

    def _remember_input(self, sig, tensor):
        """this method accumulated `tensor` as its latest sample: siblings that are handed the same memory may skip theirs"""
        import weakref
        prev = getattr(self, "_last_sig", None)
        if prev is not None and prev in _last_inputs and _last_inputs[prev][0]() is self:
            del _last_inputs[prev]
        self._last_sig = sig
        if sig is not None:
            _last_inputs[sig] = (weakref.ref(self), tensor, self.nsamples)

    def _forget_input(self):
        prev = getattr(self, "_last_sig", None)
        if prev is not None and prev in _last_inputs and _last_inputs[prev][0]() is self:
            del _last_inputs[prev]
        self._last_sig = None

    def post_batch(self):
        self._forget_input()
        leader = getattr(self, "_leader", None) or getattr(self, "_auto_leader", None)
        if leader is not None:
            assert leader.nsamples == self.nsamples, "shared Hessian: leader and follower saw different samples"
            raw = getattr(leader, "_shared_raw", None)
            if raw is not None:                                 # the leader is finished (and may have preprocessed its own copy since)
                self.H = raw.clone()
            else:                                               # leader not finished yet: finish a copy from its accumulator
                assert leader.H is not None and leader.H.dtype == torch.float64, "shared Hessian: the leader's accumulator is gone"
                self.H = (ops.hessian_finish(leader.H, leader.nsamples) if getattr(leader, "_tri", False)
                          else (leader.H / leader.nsamples).to(torch.float32))
            leader._followers = getattr(leader, "_followers", 1) - 1
            if leader._followers <= 0:
                leader._shared_raw = None
            self._leader = self._auto_leader = None
            return
        if getattr(self, "_tri", False):
            self.H = ops.hessian_finish(self.H, self.nsamples)
            self._tri = False
        else:
            self.H = (self.H / self.nsamples).to(torch.float32)
        if getattr(self, "_followers", 0) > 0:                  # followers still to come: they copy THIS tensor (preproc / fasterquant never
            self._shared_raw = self.H                           # write into it: they rebind self.H to new tensors)

    # ---- preprocessing (method.py:125-193) -------------------------------------------------------------
    def preproc(self, preproc_gptqH=False, percdamp=.01, preproc_rescale=False, preproc_proj=False,
                preproc_proj_extra=0):
        """diagonal rescale, random orthogonal projection (extra: 0 blocked butterfly + permute, 1 Kronecker,
        2 blocked without permutation), then the GPTQ dead-column / damping fix -- in that order.  W is
        written back in the layer's dtype after every stage exactly like the reference (method.py:155,179,191)."""
        self.preproc_gptqH, self.preproc_rescale, self.preproc_proj = preproc_gptqH, preproc_rescale, preproc_proj
        if OPERATOR_PREFETCH:
            pf = operator_prefetcher()
            if not preproc_proj:
                pf.drain()                                       # nothing to take: whatever was drawn ahead for this call is handed back
            pf.note_flags(preproc_proj, preproc_proj_extra)
        wdtype = self.layer.weight.data.dtype
        fused = FUSED_PREPROC and self.layer.weight.data.is_cuda and self.layer.weight.data.dim() == 2 and \
            self.layer.weight.data.dtype in (torch.float16, torch.bfloat16, torch.float32)
        if preproc_rescale:
            if fused:
                # the same operations in the same order as the torch chain below, in three launches (csrc/preproc.hip)
                # a NEW tensor, like the torch chain below and the reference (method.py:155 rebinds weight.data): the launch rewrites W in
                # place, and an alias of the old weight (a `full_W = layer.weight.data` kept for error_compute, tied weights) must not move
                w = self.layer.weight.data.clone(memory_format=torch.contiguous_format)
                H = self.H.to(torch.float32).contiguous()
                if H.data_ptr() == self.H.data_ptr():
                    H = H.clone()                                   # the torch chain leaves the caller's H untouched too
                s = ops.preproc_rescale(w, H)
                self.scaleWH = s.cpu()
                self.layer.weight.data = w
                self.H = H
            else:
                w = self.layer.weight.data.to(torch.float32)
                H = self.H.to(torch.float32)
                H = H / H.abs().max()
                diagH = torch.diag(H).clamp(min=1e-8)
                diagW2 = (w * w).sum(0).clamp(min=1e-8)             # = diag(w^T w) without the d x d product
                s = (diagH / diagW2).sqrt().sqrt().to(torch.float32).clamp(min=1e-8)
                w = w * s[None, :]
                H = (H / s[None, :]) / s[:, None]
                self.scaleWH = s.cpu()
                self.layer.weight.data = w.to(wdtype)
                self.H = H.to(torch.float32)
        if preproc_proj:
            w = self.layer.weight.data.to(torch.float32)
            H = self.H.to(torch.float32)
            gen = _GENERATORS[preproc_proj_extra]
            pre = operator_prefetcher().take(id(self), preproc_proj_extra) if OPERATOR_PREFETCH else None
            if pre is not None:
                self.projU, self.projV = pre
            else:
                self.projU = gen(w.shape[0])                     # rows first, then columns (method.py:162-163)
                self.projV = gen(w.shape[1])
            U, V = ops.OrthoOp(self.projU, w.device), ops.OrthoOp(self.projV, w.device)
            self._U, self._V = U, V
            n = H.shape[0]
            if fused:
                H = ops.preproc_trace_ridge(H.clone() if H.data_ptr() == self.H.data_ptr() else H.contiguous(), 1e-2)
            else:
                H = H * (n / (torch.trace(H) + 1e-8)) + 1e-2 * torch.eye(n, device=w.device)
            w = U.apply_cols(V.apply_rows(w))                    # U w V^T
            H = V.apply_rows(V.apply_rows(H).t().contiguous())   # V H V^T (H symmetric)
            self.layer.weight.data = w.to(wdtype)
            self.H = H.to(torch.float32)
        if preproc_gptqH:
            w = self.layer.weight.data.clone()
            H = self.H.clone()
            dead = torch.diag(H) == 0
            H[dead, dead] = 1
            w[:, dead] = 0
            idx = torch.arange(self.columns, device=self.dev)
            H[idx, idx] += percdamp * torch.mean(torch.diag(H))
            self.layer.weight.data = w.to(wdtype)
            self.H = H
        self.preproc_done = True

    def skip_operators(self, preproc_proj=True, preproc_proj_extra=0):
        """consume exactly the random draws `preproc(preproc_proj=..., preproc_proj_extra=...)` would make for this layer and throw them
        away: a rank that does NOT own this Linear (shard.block_owner_per_linear) keeps numpy's and torch's generators in step with a
        single-owner run, so the operators of the Linears it does own are the seeded ones"""
        if OPERATOR_PREFETCH:
            pf = operator_prefetcher()
            if not preproc_proj:
                pf.drain()
            pf.note_flags(preproc_proj, preproc_proj_extra)
        if not preproc_proj:
            return
        if OPERATOR_PREFETCH and operator_prefetcher().take(id(self), preproc_proj_extra) is not None:
            return
        gen = _GENERATORS[preproc_proj_extra]
        gen(self.rows)
        gen(self.columns)

    def postproc(self):
        """exact inverse of the projection, then of the rescale (method.py:195-214)."""
        assert self.preproc_done is True
        wdtype = self.layer.weight.data.dtype
        if self.preproc_proj:
            w = self.layer.weight.data.to(torch.float32)
            H = self.H.to(torch.float32)
            U, V = self._U, self._V
            w = U.apply_cols(V.apply_rows(w, transpose=True), transpose=True)                 # U^T w V
            H = V.apply_rows(V.apply_rows(H, transpose=True).t().contiguous(), transpose=True)  # V^T H V
            self.layer.weight.data = w.to(wdtype)
            self.H = H
        if self.preproc_rescale:
            s = self.scaleWH.to(self.layer.weight.device)
            w = self.layer.weight.data / s[None, :]               # fp16 tensor / fp32 tensor -> fp32
            H = (self.H * s[:, None]) * s[None, :]
            self.layer.weight.data = w.to(wdtype)
            self.H = H

    def free(self):
        if DEBUG:
            self.inp1 = self.out1 = None
        self._forget_input()
        self.H = None
        self.Losses = None
        self.Trace = None
        self.scaleWH = None
        self.projU = None
        self.projV = None
        self._U = self._V = None
        torch.cuda.empty_cache()

    def error_compute(self, full_W, quant_W):
        """proxy loss tr(dW H dW^T) and max(H) (method.py:228-233)."""
        dW = full_W.float() - quant_W.float()
        self.error = ((dW @ self.H.float()) * dW).sum().item()
        self.Hmag = self.H.max().item()
