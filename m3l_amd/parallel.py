"""Data-parallel gradient exchange for the MAE step: one process per GPU, RCCL all-reduce over xGMI.

The reference trains the MAE in a single process (ppo_mae.py:182-183,262-266); multi-GPU data parallelism is new
capability here: samples are independent, so the only exchange step is the sum-all-reduce of the
gradients (7.28 M fp32 = 29 MB for the ViT-Tiny config), averaged over ranks.

Design for MI355X: all gradients live in ONE flat fp32 buffer ordered as the backward produces them
(heads -> decoder -> enc/dec glue -> encoder -> patch embed -> learned position tables); `.grad` of every parameter is a view
into it and the HIP backward writes it in place.  Once the finished prefix of the buffer holds >= 4 MB, that slice is all-reduced
asynchronously by RCCL called DIRECTLY by the library (csrc/comm.hip) on the library's own low-priority side stream — behind the
weight gradients it waits for anyway, event-fenced against the compute stream — so the decoder's 7 MB travel while the encoder
backward is still computing and the process keeps exactly two kernel-bearing streams (a third one, e.g. torch.distributed's NCCL
stream, costs 7-48 % of the step on this stack: DESIGN.md section 6).  torch.distributed carries only the 128-byte communicator id and
is the fallback / the CPU (gloo) path.  Parameters that never receive a gradient under sincos encodings (encoder.pos_embedding,
decoder_pos_emb) are excluded (`grad is None`); with use_sincosmod_encodings=False they are trained (pretrain_models.py:218-219,
280-287) and own the trailing span of the buffer (their gradient arrives through autograd after the embed / glue backward).
"""
import os
import re

import torch
import torch.distributed as dist


def _bucket_of(name: str) -> int:
    if name.startswith(("to_pixels", "to_tactiles")):
        return 0
    if name.startswith("decoder.") and not name.startswith("decoder_"):
        return 1
    if name.startswith(("enc_to_dec", "mask_token", "decoder_modality_embedding")):
        return 2
    if name.startswith("encoder.transformer"):
        return 3
    if name in GradSync.SKIP:
        return 5    # learned position tables (use_sincosmod_encodings=False): final only when the whole backward is
    return 4        # patch embed / EarlyCNN stems, encoder modality embedding


def _layer_of(name: str) -> int:
    m = re.search(r"\.layers\.(\d+)\.", name)
    if m:
        return int(m.group(1))
    return 1 << 30 if (name.endswith("transformer.norm.weight") or name.endswith("transformer.norm.bias")
                       or name in ("decoder.norm.weight", "decoder.norm.bias")) else 0


_RCCL_RANKS = {}          # global ranks (in group order) of the process group the library's one RCCL communicator was created for


class GradSync:
    """Flat-buffer, bucketed, backward-overlapped gradient all-reduce (average)."""

    SKIP = ("encoder.pos_embedding", "decoder_pos_emb.weight")

    def __init__(self, module: torch.nn.Module, process_group=None, force_comm=False):
        # learned position tables: unused (no gradient, excluded) under sincos encodings, trained otherwise — then they get the trailing
        # span of the buffer: autograd accumulates their gradient (the batch sum of the token gradients, functional._pos_grads) into the
        # flat views after the embed / glue backward, and finish() sends the span with the tail
        learned_pos = not getattr(module, "use_sincosmod_encodings", True)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_comm: issue the collectives even at world size 1 (rehearses the RCCL stream / event path on a one-GPU box)
        self._comm = self.world > 1 or (force_comm and dist.is_initialized())
        seen, named = set(), []
        for name, p in module.named_parameters():            # named_parameters() already de-duplicates shared tensors
            if id(p) in seen or not p.requires_grad or (name in self.SKIP and not learned_pos):
                continue
            seen.add(id(p))
            named.append((name, p))
        # buckets in backward order; inside a transformer bucket: final norm first, then layers from the top down (the order the
        # backward finishes them), so that a chunk of layers is one contiguous slice of the flat buffer
        named.sort(key=lambda np_: (_bucket_of(np_[0]), -_layer_of(np_[0])))
        total = sum(p.numel() for _, p in named)
        dev = named[0][1].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_params = torch.empty(total, dtype=torch.float32, device=dev)     # parameters re-homed into one buffer too
        self.layers_per_chunk = int(os.environ.get("M3L_LAYERS_PER_CHUNK", 4))   # transformer backward chunk size for comm / compute overlap
        self.min_bucket_elems = int(os.environ.get("M3L_MIN_BUCKET_ELEMS", 1 << 20))   # 4 MB of fp32 gradients per collective
        self._ready, self._sent_end = [], 0
        self._span = {}                                        # id(param) -> (start, end) in the flat buffer
        self.buckets = []                                      # (start, end, n_params)
        off, cur, start, count = 0, None, 0, 0
        self._bucket_index = {}
        for name, p in named:
            b = _bucket_of(name)
            if cur is None:
                cur = b
            if b != cur:
                self.buckets.append([start, off, count])
                cur, start, count = b, off, 0
            with torch.no_grad():
                self.flat_params[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_params[off:off + p.numel()].view_as(p)
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self._span[id(p)] = (off, off + p.numel())
            self._bucket_index[id(p)] = len(self.buckets)
            off += p.numel()
            count += 1
        self.buckets.append([start, off, count])
        self._works = []
        self._unscaled = False                                 # a SUM all-reduce has been issued since the last scale / zero_grad()
        self._pending_scale = 1.0                              # finish(defer_scale=True): 1 / world left for the optimizer to fold in
        self._reduced = set()
        self._written = set()                                  # id(param) written directly by a backward since zero_grad()
        # the last weight gradients of every transformer backward (chunk) stay on the library's side stream past the end of the
        # backward call — they overlap the modules that follow — and are joined where the gradients are consumed: in front of each
        # all-reduce (on a helper stream, so that the compute stream never waits for them) and in finish(); the workspaces they read
        # are kept alive here until finish()
        self._defer = dev.type == "cuda"
        self._keep = []
        # Where the collectives run.  "rccl" (default on a GPU): RCCL called directly by the library on ITS side stream, behind the weight
        # gradients they wait for anyway — the process keeps exactly two kernel-bearing streams.  "c10d": torch.distributed's own NCCL
        # stream, reached through a helper stream that joins the compute and the side stream first (CPU / gloo tests, or when RCCL cannot
        # be initialised).  Measured with a stand-in kernel per bucket on one MI355X (M3L_FAKE_COMM): a third kernel-bearing stream costs
        # 7 % of the step at normal priority and 48 % at high priority; the same kernels on the side stream 0.8 %.
        self._direct = False
        self._helper = None
        if self._comm and dev.type == "cuda":
            # (a gloo group on GPU tensors is the one-GPU rehearsal of the multi-rank control flow: RCCL refuses two ranks per device)
            mode = os.environ.get("M3L_COMM", "rccl" if dist.get_backend(self.group) == "nccl" else "c10d")
            if mode == "rccl":
                self._direct = self._init_direct_rccl(dev)
            if not self._direct:
                # the helper only carries event waits: HIGH priority, its own hardware-queue class (a normal-priority helper can share a
                # hardware queue with the compute stream, and its wait for the weight gradients then stalls the compute stream: -7 %)
                self._helper = torch.cuda.Stream(device=dev, priority=int(os.environ.get("M3L_HELPER_PRIORITY", -1)))
        self._fake_comm = None
        fc = os.environ.get("M3L_FAKE_COMM", "0")
        if self._comm and dev.type == "cuda" and self.world == 1 and fc != "0":
            if fc == "side":
                from . import _lib as L
                self._fake_comm = torch.cuda.ExternalStream(L.lib().m3l_side_stream(), device=dev)
            else:
                self._fake_comm = torch.cuda.Stream(device=dev, priority=-1 if fc == "1" else 0)
        self.params = [p for _, p in named]
        # gradients that reach a parameter through autograd's accumulate (learned position tables, a second backward before zero_grad())
        # count as "received a gradient since zero_grad()" too: FlatAdamW leaves the rest alone, as torch.optim.AdamW skips `.grad is None`
        # (through a weak reference: the hooks outlive this object on the parameters and must not keep its flat buffers alive)
        import weakref
        me = weakref.ref(self)

        def _mark(q, _me=me):
            s = _me()
            if s is not None:
                s._written.add(id(q))
        self._hooks = [p.register_post_accumulate_grad_hook(_mark) for p in self.params if hasattr(p, "register_post_accumulate_grad_hook")]
        self._bucket_ids = sorted({_bucket_of(n) for n, _ in named})
        # direct mode: the module's autograd Functions write straight into the flat buffer and call bucket_done()
        if hasattr(module, "_sinks"):
            module._sinks.update(heads=(self, 0), glue=(self, 2), embed=(self, 4))
            module.decoder._sink = (self, 1)
            module.encoder.transformer._sink = (self, 3)

    def _init_direct_rccl(self, dev) -> bool:
        """One RCCL communicator for this process, created through the library (m3l_comm_init); the 128-byte id travels over the
        torch.distributed group that already exists.  False (with a warning) if RCCL is not available: the c10d path takes over."""
        import ctypes as C
        import warnings
        from . import _lib as L
        lib = L.lib()
        rank = dist.get_rank(self.group)

        def all_agree(ok: bool) -> bool:
            flag = torch.tensor([1.0 if ok else 0.0], device=dev)
            if self.world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return float(flag.item()) >= 1.0
        # (1) every rank probes RCCL locally and the ranks agree BEFORE anyone enters the collective ncclCommInitRank: a rank that
        # cannot load RCCL would otherwise leave the others blocked inside it
        if not all_agree(lib.m3l_comm_available() == 0):
            warnings.warn("m3l_amd: RCCL not available to the library on every rank (%s); gradient all-reduce through torch.distributed" % L.last_error())
            return False
        ranks = tuple(dist.get_process_group_ranks(self.group)) if self.group is not None else tuple(range(dist.get_world_size()))
        if all_agree(lib.m3l_comm_world() == self.world):      # a live communicator of this process (an earlier GradSync): shared ...
            # ... but only by the SAME set of ranks in the same order (a second GradSync on another process group of equal size
            # would all-reduce over the wrong ranks without an error)
            if not all_agree(_RCCL_RANKS.get("ranks") == ranks):
                raise RuntimeError(f"m3l_amd: the library's RCCL communicator was created for ranks {_RCCL_RANKS.get('ranks')}, this "
                                   f"GradSync's process group is {ranks}; one communicator per process — set M3L_COMM=c10d for this group")
            return True
        ident = C.create_string_buffer(128)
        ok = 1
        if rank == 0:
            ok = 1 if lib.m3l_comm_unique_id(ident) == 0 else 0
        box = [bytes(ident.raw) if ok else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        if box[0] is None:
            warnings.warn("m3l_amd: ncclGetUniqueId failed on rank 0 (%s); gradient all-reduce through torch.distributed" % L.last_error())
            return False
        with torch.cuda.device(dev):
            rc = lib.m3l_comm_init(C.create_string_buffer(box[0], 128), rank, self.world)
        # (2) ncclCommInitRank either succeeds on every rank or the job cannot go on: a rank that failed here has left its peers with a
        # communicator that will never complete a collective, so there is no safe fallback — every rank raises (the process exits non-zero)
        if not all_agree(rc == 0):
            lib.m3l_comm_destroy()
            raise RuntimeError("m3l_amd: m3l_comm_init failed on at least one rank (%s): the ranks disagree about the RCCL communicator; "
                               "set M3L_COMM=c10d to run the gradient all-reduce through torch.distributed" % L.last_error())
        # (3) direct evidence that the communicator spans the ranks: sum of (rank + 1) over the library's own all-reduce
        probe = torch.tensor([float(rank + 1)], device=dev)
        L.check(lib.m3l_comm_allreduce(probe.data_ptr(), 1, torch.cuda.current_stream().cuda_stream), "m3l_comm_allreduce (probe)")
        L.check(lib.m3l_side_join(torch.cuda.current_stream().cuda_stream), "m3l_side_join")
        want = self.world * (self.world + 1) / 2
        if float(probe.item()) != want:
            raise RuntimeError(f"m3l_amd: RCCL probe all-reduce returned {float(probe.item())}, expected {want} for {self.world} ranks")
        self.rccl_probe = float(probe.item())
        _RCCL_RANKS["ranks"] = ranks
        return True

    def __del__(self):
        for h in getattr(self, "_hooks", ()):
            try:
                h.remove()
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass

    def zero_grad(self):
        self.flat.zero_()
        self._unscaled = False
        self._pending_scale = 1.0
        self._reduced = set()
        self._written = set()
        self._ready, self._sent_end = [], 0

    # The flat buffer is laid out in the order the backward finishes (heads -> decoder top-down -> glue -> encoder top-down ->
    # embed), so finished spans grow a contiguous prefix.  A collective is issued for the finished-but-unsent prefix once it
    # holds at least `min_bucket_elems` elements (few, large all-reduces: xGMI rings are latency-bound below a few MB), and
    # finish() sends what is left.
    def _span_ready(self, s, e, flush=False):
        self._ready.append((s, e))
        end, grew = self._sent_end, True
        while grew:
            grew = False
            for a, b in self._ready:
                if a <= end < b:
                    end, grew = b, True
        if end > self._sent_end and (flush or end - self._sent_end >= self.min_bucket_elems or end == self.flat.numel()):
            self._unscaled = True
            if self._direct:
                # in place, on the library's side stream: behind the compute stream's work so far (event) and behind the weight gradients
                # already queued there; finish() joins the side stream before the optimizer reads the sums
                from . import _lib as L
                seg = self.flat[self._sent_end:end]
                L.check(L.lib().m3l_comm_allreduce(seg.data_ptr(), seg.numel(), torch.cuda.current_stream().cuda_stream), "m3l_comm_allreduce")
                if self._fake_comm is not None:
                    with torch.cuda.stream(self._fake_comm):
                        seg.mul_(1.0)
                self._sent_end = end
                return
            if self._helper is not None:
                # the collective must see the compute stream's gradients AND the weight gradients still on the library's side stream:
                # both are joined onto a helper stream, the all-reduce is issued from there, the compute stream runs on
                from . import _lib as L
                self._helper.wait_stream(torch.cuda.current_stream())
                L.check(L.lib().m3l_side_join(self._helper.cuda_stream), "m3l_side_join")
                with torch.cuda.stream(self._helper):
                    work = dist.all_reduce(self.flat[self._sent_end:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    if self._fake_comm is not None:
                        # rehearsal only (M3L_FAKE_COMM=1 at world size 1, where RCCL launches nothing): a kernel of the collective's size
                        # on a third, high-priority stream, ordered like c10d orders its own stream
                        self._fake_comm.wait_stream(self._helper)
                        with torch.cuda.stream(self._fake_comm):
                            self.flat[self._sent_end:end].mul_(1.0)
                        self._helper.wait_stream(self._fake_comm)
            else:
                work = dist.all_reduce(self.flat[self._sent_end:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append(work)
            self._sent_end = end

    def bucket_done(self, bucket_id: int):
        """Called from the backward of the module that owns `bucket_id` once its gradients are in the flat buffer."""
        self._reduced.add(bucket_id)
        if not self._comm or bucket_id not in self._bucket_ids:
            return
        s, e, _ = self.buckets[self._bucket_ids.index(bucket_id)]
        self._span_ready(s, e)

    def range_done(self, bucket_id: int, params, last: bool):
        """Part of a bucket is final (a chunk of transformer layers): that contiguous slice may travel."""
        spans = [self._span[id(p)] for p in params if p is not None and id(p) in self._span]
        if last:
            self._reduced.add(bucket_id)
        if not self._comm or not spans:
            return
        self._span_ready(min(a for a, _ in spans), max(b for _, b in spans))

    def finish(self, defer_scale: bool = False):
        """Send whatever has not travelled yet (buckets nobody reported: modules whose gradients come from several Functions),
        then make the compute stream wait for every outstanding all-reduce (call before optimizer.step()).  Idempotent: the 1 / world
        scale is applied once per set of collectives.  defer_scale=True leaves the sums in the buffer and hands 1 / world to the
        optimizer (FlatAdam.step folds it into its one launch: take_scale())."""
        if self._comm:
            for b in self._bucket_ids:
                if b not in self._reduced:
                    self.bucket_done(b)
            if self._sent_end < self.flat.numel():
                self._span_ready(self._sent_end, self.flat.numel(), flush=True)
        if self._defer:
            from . import _lib as L
            L.check(L.lib().m3l_side_join(torch.cuda.current_stream().cuda_stream), "m3l_side_join")
            self._keep.clear()
        for w in self._works:
            w.wait()
        self._works = []
        if self._unscaled and self.world > 1:      # SUM then scale: works on every backend (gloo has no AVG)
            if defer_scale:
                self._pending_scale *= 1.0 / self.world
            else:
                self.flat.mul_(1.0 / self.world)
        self._unscaled = False
        if not defer_scale and self._pending_scale != 1.0:     # an earlier deferred scale nobody consumed: apply it now
            self.flat.mul_(self._pending_scale)
            self._pending_scale = 1.0

    def take_scale(self) -> float:
        """The factor a finish(defer_scale=True) left for the optimizer (1.0 otherwise); consumed by the call."""
        s, self._pending_scale = self._pending_scale, 1.0
        return s

    def close(self):
        """Tear the library's RCCL communicator down (collective: every rank calls it, before destroy_process_group())."""
        if self._direct:
            from . import _lib as L
            torch.cuda.synchronize()
            L.lib().m3l_comm_destroy()
            self._direct = False
            self._comm = False

    def reduce_now(self):
        """Non-overlapped variant (used by tests / when hooks are not wanted)."""
        if self._comm:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)


class FlatAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics (the reference's MAE optimizer, ppo_mae.py:182-183)
    as ONE HIP launch over the GradSync's flat parameter / gradient buffers (m3l_adam_step)."""

    def __init__(self, sync: GradSync, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        self.sync, self.lr, self.betas, self.eps, self.weight_decay = sync, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(sync.flat)
        self.exp_avg_sq = torch.zeros_like(sync.flat)
        self.step_count = 0
        # capturable (as torch.optim.Adam(capturable=True)): the step counter lives on the device, so step() can be recorded
        # into a HIP graph and replayed
        self.capturable = capturable
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=sync.flat.device) if capturable else None
        self._bc_dev = torch.zeros(2, dtype=torch.float32, device=sync.flat.device) if capturable else None
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "params": sync.params}]

    def zero_grad(self, set_to_none: bool = False):
        self.sync.zero_grad()

    def step(self):
        from . import _lib as L
        if self.sync._keep or self.sync._unscaled:     # a backward whose side-stream work / collectives have not been joined yet
            self.sync.finish(defer_scale=not self.capturable)
        gscale = 1.0
        if not self.capturable:
            gscale = self.sync.take_scale()
        elif self.sync._pending_scale != 1.0:
            self.sync.finish()
        g = self.param_groups[0]
        if self.capturable:
            L.check(L.lib().m3l_adam_step_dev(self.sync.flat_params.data_ptr(), self.sync.flat.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self.sync.flat.numel(), g["lr"], g["betas"][0], g["betas"][1],
                                              g["eps"], g["weight_decay"], self._step_dev.data_ptr(), self._bc_dev.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "m3l_adam_step_dev")
            return
        self.step_count += 1
        L.check(L.lib().m3l_adam_step_scaled(self.sync.flat_params.data_ptr(), self.sync.flat.data_ptr(), self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.sync.flat.numel(), g["lr"], g["betas"][0], g["betas"][1],
                                             g["eps"], g["weight_decay"], self.step_count, gscale, torch.cuda.current_stream().cuda_stream),
                "m3l_adam_step_scaled")

    def state_dict(self):
        return {"step": int(self._step_dev.item()) if self.capturable else self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        if self.capturable:
            self._step_dev.fill_(self.step_count)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class FlatAdamW(FlatAdam):
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay=1e-2) — the optimizer of VTMAE.initialize_training
    (pretrain_models.py:675) — with the `clip_grad_norm_(parameters, max_grad_norm)` of VTMAE.train_iterations
    (pretrain_models.py:710) fused in: one reduction over the flat gradient buffer + one update launch (m3l_adamw_step) instead of
    torch's per-tensor norm / multiply / AdamW kernels.  max_grad_norm=None: no clipping.  After step(), `last_grad_norm` is a device
    scalar holding the gradient norm clip_grad_norm_ would have returned; the clipped gradients are left in `.grad`, as it leaves them."""

    def __init__(self, sync: GradSync, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        super().__init__(sync, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=False)
        self.max_grad_norm = max_grad_norm
        self._norm_ws = torch.zeros(1026, dtype=torch.float32, device=sync.flat.device)
        self.last_grad_norm = self._norm_ws[1025]

    def step(self):
        from . import _lib as L
        if self.sync._keep or self.sync._unscaled:
            self.sync.finish(defer_scale=True)
        gscale = self.sync.take_scale()
        g = self.param_groups[0]
        self.step_count += 1
        mx = float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0
        # torch.optim.AdamW (the reference's optimizer, pretrain_models.py:675) skips parameters whose .grad is None: under
        # train_iterations(no_tactile=True), or with the unused patch-embed tensors of early_conv_masking=True, those parameters are neither
        # decayed nor given moments.  Here every .grad is a (zeroed) view of the flat buffer, so the one launch over the whole buffer would
        # decay them: parameters that received no gradient since zero_grad() are put back afterwards (their zero gradients do not change the
        # norm).  [A parameter that receives gradients in some steps only keeps the global step count in its bias correction; torch counts
        # its own steps.]
        idle = [q for q in self.sync.params if id(q) not in self.sync._written]
        saved = []
        if idle and len(idle) < len(self.sync.params):
            for q in idle:
                a, b = self.sync._span[id(q)]
                saved.append((a, b, self.sync.flat_params[a:b].clone(), self.exp_avg[a:b].clone(), self.exp_avg_sq[a:b].clone()))
        L.check(L.lib().m3l_adamw_step(self.sync.flat_params.data_ptr(), self.sync.flat.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.sync.flat.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                       g["weight_decay"], self.step_count, gscale, mx, self._norm_ws.data_ptr(), 1,
                                       torch.cuda.current_stream().cuda_stream), "m3l_adamw_step")
        for a, b, pv, mv, vv in saved:
            self.sync.flat_params[a:b].copy_(pv)
            self.exp_avg[a:b].copy_(mv)
            self.exp_avg_sq[a:b].copy_(vv)
