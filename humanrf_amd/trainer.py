"""Training step of HumanRF on the gfx950 kernels.

`TrainEngine.train_iteration()` is the loop body of Trainer.train (humanrf/trainer.py:135-187):
  - batch growing: draw rays, sample, prune, until >= 0.9 * samples_max_batch_size (trainer.py:138-163)
  - merge_input_batches(max = 1.1 * samples_max_batch_size)                      (trainer.py:170-172)
  - train_step: random background, render, Huber + BCE loss, backward, Adam, LR   (trainer.py:229-255)
It calls the kernels directly instead of going through autograd: the backward is
loss -> composite_bwd -> mlp_bwd -> encode4d_bwd -> fused Adam. Loss scaling is the reference's: a device-resident
GradScaler (init 65536, backoff / growth applied by the optimizer launch, found_inf -> the step is skipped,
trainer.py:74,250-252) times tcnn's internal loss_scale of 128. The modules in humanrf_amd.scene_representation /
volume_rendering expose the same math through autograd for callers that keep the reference's own Trainer.

Multi-GPU: one process per GPU, rays sharded (each rank owns its pool slice and RNG stream), tables
replicated. Gradient exchange over RCCL before the optimizer, default `exchange="sharded"`: reduce-scatter of the table
gradients of the segments in the pools, Adam on the 1/N of every segment a rank owns, all-gather of the fp16 tables (issued
behind the optimizer, waited for by the next reader of the tables: the next step's prune march; the next step's sampler
stages run under it); vectors / MLPs / embeddings / flags in one small all-reduce. `exchange="allreduce"`: one all-reduce of
the table gradients instead, every rank steps everything."""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import ops
from .dataset.input_batch import InputBatch
from .fast_path import StepCollector
from .input import merge_input_batches
from .scene_representation.humanrf import HumanRF
from .volume_rendering import prune_samples


def allreduce_gradients(flat_grads: torch.Tensor, big_numel: int, world_size: int, group=None,
                        transport_dtype: Optional[torch.dtype] = torch.float32, wire: Optional[torch.Tensor] = None,
                        average: bool = True, head: bool = True, tail: bool = True, wait: bool = True,
                        head_ranges: Optional[Sequence[Tuple[int, int]]] = None, force: bool = False):
    """Reduce the flat gradient buffer over the data-parallel group, in place (mean, or sum with average=False --
    the training engine folds 1/world into the optimizer's unscale factor and saves a pass over the buffer).
    The first `big_numel` elements ("head": the hash tables, 10^7..10^8 values) travel in `transport_dtype`: fp32 by
    default (the N-rank gradient is then the exact fp32 mean of the per-rank gradients); bf16 is opt-in -- it halves the
    bytes every xGMI link carries (fp32 exponent range, so the scaled gradients need no re-scaling) at an 8-bit mantissa
    per summand, tests/test_gpu_data_parallel.py states the resulting bound. The "tail" (vectors, MLP weights,
    embeddings, found_inf flag, touched-segment flags) always travels in fp32.
    head_ranges: [(start, end)) element ranges of the head that can hold gradients on ANY rank (the tables of the
    temporal segments whose frames are in the pools); everything outside is zero on every rank and is
    not exchanged. None = the whole head. `wire`: caller-owned transport buffer (big_numel, transport_dtype).
    head / tail select which part to exchange; wait=False returns a callable that completes the exchange (so that
    further work can be enqueued under it). No-op for world_size == 1 unless `force` (a one-rank group still issues the
    collective calls: how the RCCL code path is exercised on a one-GPU box)."""
    if world_size <= 1 and not force:
        return (lambda: None) if not wait else None
    import torch.distributed as dist
    inv = 1.0 / world_size
    big, small = flat_grads[:big_numel], flat_grads[big_numel:]
    ranges = [(0, big_numel)] if head_ranges is None else [(int(a), int(b)) for a, b in head_ranges if b > a]
    fp32_wire = transport_dtype is None or transport_dtype == torch.float32
    handles = []
    if head:
        if not fp32_wire and wire is None:
            wire = torch.empty(big_numel, dtype=transport_dtype, device=flat_grads.device)
        for a, b in ranges:
            if fp32_wire:
                handles.append(dist.all_reduce(big[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True))
            else:
                wire[a:b].copy_(big[a:b])  # one fused cast pass, no fp32 temporary
                handles.append(dist.all_reduce(wire[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True))
    if tail:
        handles.append(dist.all_reduce(small, op=dist.ReduceOp.SUM, group=group, async_op=True))

    def finish():
        for h in handles:
            h.wait()
        if head:
            for a, b in ranges:
                if not fp32_wire:
                    big[a:b].copy_(wire[a:b])
                if average:
                    big[a:b].mul_(inv)
        if average and tail:
            small.mul_(inv)

    if not wait:
        return finish
    finish()
    return None


class TableShardExchange:
    """The table half of the data-parallel step as SURVEY.md 8(e) lays it out for 8 fully connected GPUs: reduce-scatter of the
    gradients, Adam on the shard a rank owns, all-gather of the fp16 tables. Segment s's four tables occupy
    [table_ranges[s]) of the flat parameter / gradient buffers; rank r owns the r-th of `world_size` equal slices of every
    segment (so that whichever segments a step touches, the optimizer work and the link traffic are balanced)."""

    def __init__(self, table_ranges: Sequence[Tuple[int, int]], world_size: int, rank: int, group=None):
        self.world_size, self.rank, self.group = int(world_size), int(rank), group
        self.table_ranges = [(int(a), int(b)) for a, b in table_ranges]
        self.own_ranges = []
        for sidx, (a, b) in enumerate(self.table_ranges):
            # entries per encoding are a multiple of 8 (tcnn pads every level): (b - a) = 8 e divides by 1, 2, 4, 8 ranks
            if (b - a) % self.world_size:
                raise ValueError(f"segment {sidx}: {b - a} table values do not divide over {self.world_size} ranks")
            sz = (b - a) // self.world_size
            self.own_ranges.append((a + self.rank * sz, a + (self.rank + 1) * sz))
        # reduce_scatter_tensor / all_gather_into_tensor are what RCCL runs; gloo (the CPU tests, and two test ranks on one
        # GPU) has neither for these tensors and gets all_reduce / the list form of all_gather instead. Chosen by the
        # backend, never by catching an error: a failing RCCL collective must surface.
        self.collectives_used = set()         # names of the torch.distributed calls that actually ran (bench.py reports them)
        self.coalesce = True                  # several segments of one phase travel as one backend launch (see _coalesced)
        self.bytes_issued = 0                 # payload handed to the table collectives since the caller last cleared it (per rank)
        self.issue_log = []                   # (phase, segments) in issue order since the caller last cleared it

    @property
    def tensor_collectives(self) -> bool:
        import torch.distributed as dist
        return str(dist.get_backend(self.group)) != "gloo"

    def self_check(self, device, numel: int = 1 << 18, _dist=None) -> None:
        """Start-up probe of the two collectives the sharded exchange rests on, in the IN-PLACE forms it uses them in (the output
        of reduce_scatter_tensor is the rank-th slice of its own input; the input of all_gather_into_tensor is the rank-th slice
        of its own output): on a 1 MB buffer of small integers (sums are exact in any order) the pair must reproduce all_reduce.
        Raises on a mismatch -- a backend that does not define the in-place forms the way NCCL / RCCL do must not train silently
        on wrong gradients -- and raises on EVERY rank when any rank saw one (the verdicts are max-reduced first: a rank that
        raised alone would leave the others waiting in their next collective). A no-op on gloo (which runs the all_reduce / list
        all_gather stand-ins). `_dist`: a stand-in for torch.distributed (tests)."""
        if _dist is None:
            import torch.distributed as _dist
            if not self.tensor_collectives:
                return
        dist = _dist
        w, r = self.world_size, self.rank
        n = (numel // max(w, 1)) * max(w, 1)
        base = (torch.arange(n, device=device, dtype=torch.float32) * 7.0) % 61.0 - 30.0        # integers in [-30, 30]
        mine = base * float(r + 1) + float(r)
        ref = mine.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=self.group)
        got = mine.clone()
        sz = n // w
        dist.reduce_scatter_tensor(got[r * sz:(r + 1) * sz], got, op=dist.ReduceOp.SUM, group=self.group)
        bad_rs = not torch.equal(got[r * sz:(r + 1) * sz], ref[r * sz:(r + 1) * sz])
        got[r * sz:(r + 1) * sz].copy_(ref[r * sz:(r + 1) * sz])     # (the second probe starts from the right slice either way)
        dist.all_gather_into_tensor(got, got[r * sz:(r + 1) * sz], group=self.group)
        bad_ag = not torch.equal(got, ref)
        # the coalesced forms the exchange issues when a group holds several segments (_coalesced): two ranges in one backend launch.
        # A backend whose coalescing path misbehaves -- or does not exist -- only loses the coalescing (by consensus), not the run.
        bad_co = False
        if self.coalesce and hasattr(dist, "_coalescing_manager") and n >= 4 * w:
            try:
                n2 = (n // 2 // w) * w
                ranges = [(0, n2), (n2, n)]
                got = mine.clone()
                with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                    for a, b in ranges:
                        s2 = (b - a) // w
                        dist.reduce_scatter_tensor(got[a + r * s2:a + (r + 1) * s2], got[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True)
                cm.wait()
                for a, b in ranges:
                    s2 = (b - a) // w
                    bad_co |= not torch.equal(got[a + r * s2:a + (r + 1) * s2], ref[a + r * s2:a + (r + 1) * s2])
                    got[a + r * s2:a + (r + 1) * s2].copy_(ref[a + r * s2:a + (r + 1) * s2])
                with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                    for a, b in ranges:
                        s2 = (b - a) // w
                        dist.all_gather_into_tensor(got[a:b], got[a + r * s2:a + (r + 1) * s2], group=self.group, async_op=True)
                cm.wait()
                bad_co |= not torch.equal(got, ref)
            except Exception:      # (raised before anything is launched, and on every rank alike: same library, same arguments)
                bad_co = True
        verdict = torch.tensor([float(bad_rs), float(bad_ag), float(bad_co)], device=device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX, group=self.group)
        any_rs, any_ag, any_co = (bool(v) for v in verdict.tolist())
        if any_co:
            self.coalesce = False
            self.collectives_used.add("coalesced launches refused by the start-up probe: one collective per segment")
        if any_rs:
            raise RuntimeError("TableShardExchange.self_check: in-place reduce_scatter_tensor does not deliver a rank's slice of "
                               f"the all-reduced buffer (this rank: {'wrong' if bad_rs else 'right'})")
        if any_ag:
            raise RuntimeError("TableShardExchange.self_check: in-place all_gather_into_tensor does not reproduce the all-reduced "
                               f"buffer from the ranks' slices (this rank: {'wrong' if bad_ag else 'right'})")
        self.collectives_used.add("self_check: reduce_scatter_tensor + all_gather_into_tensor == all_reduce (1 MB probe)")

    def _coalesced(self, n_ops: int):
        """Several tensor collectives of one phase as ONE backend launch (ncclGroupStart / End through torch's coalescing manager:
        at 50 frames a step exchanges 7 segments, at 1 000 frames up to 142 -- one RCCL launch per phase instead of one per
        segment). Only where torch has the fast path (reduce_scatter_tensor / all_gather_into_tensor on RCCL) and there is
        more than one operation to put together."""
        import contextlib
        import torch.distributed as dist
        if self.coalesce and self.tensor_collectives and n_ops > 1 and hasattr(dist, "_coalescing_manager"):
            return dist._coalescing_manager(group=self.group, async_ops=True)
        return contextlib.nullcontext(None)

    def reduce_scatter(self, grads: torch.Tensor, segments: Sequence[int]):
        """Start the reduce-scatter (sum over ranks) of the table gradients of `segments`: this rank's shard of every
        segment lands in place inside `grads` -> callable that waits for it. The collective is ordered behind what the current
        stream holds NOW (the accumulate launch of these segments), not behind what is enqueued afterwards."""
        import torch.distributed as dist
        handles = []
        with self._coalesced(len(segments)) as cm:
            for sidx in segments:
                (a, b), (oa, ob) = self.table_ranges[sidx], self.own_ranges[sidx]
                self.bytes_issued += (b - a) * grads.element_size()
                if self.tensor_collectives:
                    # in place: the output is the rank-th slice of the input (what NCCL / RCCL define as in-place reduce-scatter)
                    h = dist.reduce_scatter_tensor(grads[oa:ob], grads[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    self.collectives_used.add("reduce_scatter_tensor")
                else:
                    h = dist.all_reduce(grads[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    self.collectives_used.add("all_reduce (gloo stand-in for reduce_scatter_tensor)")
                if cm is None:
                    handles.append(h)
        if cm is not None:
            handles.append(cm)
            self.collectives_used.add("coalesced launches (torch.distributed._coalescing_manager)")
        self.issue_log.append(("reduce_scatter", tuple(int(x) for x in segments)))

        def finish():
            for h in handles:
                h.wait()
        return finish

    def all_gather(self, tensor: torch.Tensor, segments: Sequence[int], wait: bool = True):
        """Every rank's shard of `tensor` (indexed like the flat table buffer) -> all ranks, in place, one collective per
        segment. wait=False -> callable that makes the current stream wait for them (so that work which does not read
        `tensor` can be enqueued under the exchange)."""
        import torch.distributed as dist
        handles = []
        with self._coalesced(len(segments)) as cm:
            for sidx in segments:
                (a, b), (oa, ob) = self.table_ranges[sidx], self.own_ranges[sidx]
                whole, mine = tensor[a:b], tensor[oa:ob]
                self.bytes_issued += (b - a) * tensor.element_size()
                if self.tensor_collectives:
                    h = dist.all_gather_into_tensor(whole, mine, group=self.group, async_op=True)
                    self.collectives_used.add("all_gather_into_tensor")
                    if cm is None:
                        handles.append(h)
                else:
                    parts = list(whole.view(self.world_size, -1).unbind(0))
                    dist.all_gather(parts, mine.clone(), group=self.group)      # list form: every backend has it
                    self.collectives_used.add("all_gather (gloo stand-in for all_gather_into_tensor)")
        if cm is not None:
            handles.append(cm)
            self.collectives_used.add("coalesced launches (torch.distributed._coalescing_manager)")
        self.issue_log.append(("all_gather", tuple(int(x) for x in segments)))

        def finish():
            for h in handles:
                h.wait()
        if not wait:
            return finish
        finish()
        return None


@dataclass
class StepStats:
    num_rays: int = 0            # rays entering train_step (post mask, post merge)
    num_rays_drawn: int = 0      # R0 summed over the batch-growing iterations
    num_samples_pre: int = 0     # N0: occupancy-surviving samples handed to the pruning pass (int, or a device scalar)
    num_samples: int = 0         # N1: samples entering render
    sums: torch.Tensor = None    # device (3,): sum huber, sum bce, sum squared error


class TrainEngine:
    def __init__(self, model: HumanRF, loader, lr: float = 1e-2, lr_decay: float = 0.5, max_steps: int = 50_001,
                 samples_max_batch_size: int = 640_000, rays_initial_batch_size: int = 8192,
                 bce_loss_weight: float = 1e-3, huber_delta: float = 0.01, grad_scale: float = 65536.0,
                 scaler_growth_interval: int = 100_000, internal_grad_scale: float = 128.0,
                 world_size: int = 1, process_group=None, transport_dtype=torch.float32, fast_collect: bool = True,
                 exchange_touched_only: bool = True, pipeline_pieces: int = 1, table_scatter: str = "auto",
                 exchange: str = "sharded", rank: Optional[int] = None, force_collectives: bool = False,
                 gradient_boundaries: str = "fp16", overlap_vector_scatter: bool = True, mlp_backward: str = "fused",
                 exchange_groups: int = 4):
        self.model, self.loader = model, loader
        self.lr0, self.lr_decay, self.max_steps = lr, lr_decay, max_steps
        self.samples_max = samples_max_batch_size
        self.rays_initial = rays_initial_batch_size
        self.bce_w, self.delta = bce_loss_weight, huber_delta
        # Loss scaling as the reference has it: torch.cuda.amp.GradScaler (trainer.py:74: default init_scale 65536,
        # growth_interval = config.training.scaler_growth_interval, example_humanrf.py:22: 100000; backoff 0.5 on a
        # non-finite gradient) times tcnn's internal loss_scale of 128 for half-precision modules (its torch binding
        # multiplies dL/dy by it before the fp16 backward and divides the results). `grad_scale` is the GradScaler's
        # init_scale; the scaler itself lives on the device (ops.grad_scaler) and is updated by the optimizer kernel.
        self.internal_grad_scale = float(internal_grad_scale)
        # "fp16" (default): the gradient is rounded through half wherever the reference's modules hand each other half tensors at
        # the GradScaler's scale (include/hrf.h, grad_boundary): contributions below ~9e-13 vanish as they do in the reference,
        # and the entries they alone touch do not move -- on one batch 0.003 % of the moved table entries move on one side only
        # (reference's Trainer.train_step over the drop-in modules vs this engine). "fp32": the fused backward keeps fp32 from the
        # loss to the tables, contributions of any size reach Adam and ~10 % more entries take a first step of lr. Same speed;
        # novel-view PSNR at camera_embedding_dim 0 after 2 080 steps: 34.69 dB ("fp16", mean of four runs) vs 34.82 dB ("fp32")
        # (the example configuration's embedding dimension 2 is measured there separately).
        if gradient_boundaries not in ("fp32", "fp16"):
            raise ValueError("gradient_boundaries must be 'fp32' or 'fp16'")
        self.gradient_boundaries = gradient_boundaries
        self._gb = self.internal_grad_scale if gradient_boundaries == "fp16" else 0.0
        # measurement aids (tools/psnr_variance.py): the boundary applied by the MLP backward (dL/d(sigma_net output), dL/d(features))
        # and the one applied by the table scatter (the compose op's four per-encoding gradients) can be switched separately
        self._gb_mlp = self._gb_tables = self._gb
        self.world_size, self.group, self.transport_dtype = world_size, process_group, transport_dtype
        self.exchange_touched_only = exchange_touched_only
        # Data-parallel exchange of the table gradients: "sharded" = reduce-scatter, every rank runs Adam on
        # its 1/N of each segment's tables, all-gather of the fp16 tables the kernels read (1.5 S (N-1)/N bytes per rank and
        # step over the xGMI links instead of the all-reduce's 2 S (N-1)/N, and 1/N of the optimizer's HBM traffic);
        # "allreduce" = every rank steps everything. Vectors, MLPs, embeddings and the flags are all-reduced either way.
        if exchange not in ("sharded", "allreduce"):
            raise ValueError("exchange must be 'sharded' or 'allreduce'")
        # force_collectives: run the data-parallel step (every torch.distributed call of it) on a group of ONE rank -- the
        # collectives degenerate to copies but are real RCCL calls on RCCL's streams: how the N > 1 code path is executed on
        # a one-GPU box (bench.py --force-collectives). Needs an initialised process group.
        self.force_collectives = bool(force_collectives)
        self.exchange = exchange if self._dp else "allreduce"
        # Data parallel, binned scatter: the table gradients are accumulated and handed to the exchange in up to this many groups
        # of temporal segments, each group's collective starting while the next group is still being accumulated and the vector
        # half of the backward has not begun (train_step). 1 = accumulate everything, then exchange (rounds 3-5).
        self.exchange_groups = max(1, int(exchange_groups))
        # Data parallel, binned scatter: ONE accumulate launch that signals each segment group's completion to the stream its collective
        # is issued from (hrf_scatter_accumulate_signalled + hipStreamWaitValue64) instead of one accumulate launch per group -- the
        # groups' collectives still start as their gradients complete, and the scatter keeps the cost of a single launch (four launches:
        # +0.2 ms of tails, profiles/r06_dp_lines.txt). False, or a device without stream value waits: a launch per group.
        self.exchange_signalled = True
        self._sig_stream = None
        self._group_done = None               # int64 counters on the device, cumulative over the steps
        self._group_goal = [0] * 8            # what each counter reaches once the current step's launch is through with the group
        self.exchange_bytes = 0        # table-gradient payload handed to the exchange in the last step (this rank)
        self.exchange_issue_log = []   # (phase, segments) of the last step's table collectives in issue order
        self.exchange_issue_log_mode = None   # how the last step's accumulate launches and collectives were interleaved
        if self.exchange == "sharded" and transport_dtype not in (None, torch.float32):
            raise ValueError("the sharded exchange reduces in fp32 (use exchange='allreduce' for a bf16 wire)")
        if rank is None:
            rank = 0
            if self._dp:
                import torch.distributed as dist
                rank = dist.get_rank(process_group)
        self.rank = int(rank)
        self.betas, self.eps = (0.9, 0.99), 1e-15  # humanrf/run.py:101
        self.step = 0         # train_step calls
        self.sched_step = 0   # lr_scheduler.step() calls (LambdaLR, run.py:102-104)
        dev = model.table_params.device
        m = model
        self._params: List[torch.Tensor] = [m.table_params, m.vectors, m.sigma_params, m.color_params]
        if m.camera_embedding_dim > 0:
            self._params.append(m.camera_embeddings.weight)
        sizes = [p.numel() for p in self._params]
        self._big = sizes[0]
        self._wire = None
        if self._dp and transport_dtype not in (None, torch.float32):
            self._wire = torch.empty(self._big, dtype=transport_dtype, device=dev)
        total = sum(sizes)
        G = 1 + m.num_segments   # optimizer groups: 0 = MLPs + embeddings, 1 + s = tables and vectors of segment s
        self.num_groups = G
        # one flat fp32 gradient buffer; behind the gradients: found_inf flag and the touched flags of the groups, as
        # floats, so that ONE small all-reduce carries them together with the vector / MLP gradients
        self.flat_grad = torch.zeros(total + 1 + G, dtype=torch.float32, device=dev)
        self._grads, off = [], 0
        for n in sizes:
            self._grads.append(self.flat_grad[off:off + n])
            off += n
        self._flag_f = self.flat_grad[total:total + 1 + G]
        self.exp_avg = [torch.zeros_like(p, dtype=torch.float32) for p in self._params]
        self.exp_avg_sq = [torch.zeros_like(p, dtype=torch.float32) for p in self._params]
        self._arena = ops.Arena()
        # optimizer state on the device (hrf_adam_multi): found_inf, skipped, internal, -, steps[G], touched[G]
        self.opt_state = torch.zeros(4 + 2 * G, dtype=torch.int32, device=dev)
        self.flags = self.opt_state[0:1]
        self._touched = self.opt_state[4 + G:4 + 2 * G]
        self._skipped_seen = 0
        # pieces a step's batch is fed in on one GPU (1 = one pass, 2 or 4 = pipelined over two streams), see train_step.
        # Default 1: measured on MI355X (bench.py --ab-pieces 1,2,4, alternating in one process, 640 k samples / step):
        # 4.94 / 4.95-5.13 / 5.21-5.27 ms per step -- the scatter under the next piece's forward slows both down by what the
        # overlap gains (the forward gather needs the wave slots the scatter's wavefronts hold).
        self._pipeline_pieces = int(pipeline_pieces)
        self._batch_sorted = False
        # one GPU: run the vector-gradient scatter on a second stream under the table-gradient scatter (train_step)
        self.overlap_vector_scatter = bool(overlap_vector_scatter)
        self.vector_scatter_under = "emit"       # "emit": the vector kernel starts with the table scatter; "accumulate": behind its emit half
        # backward of the two MLPs: "fused" = hrf_mlp_bwd (one kernel, one wavefront per SIMD), "split" = hrf_color_mlp_bwd +
        # hrf_density_mlp_bwd (d_h travels through memory: 64 B per sample each way; two wavefronts per SIMD each)
        if mlp_backward not in ("fused", "split"):
            raise ValueError("mlp_backward must be 'fused' or 'split'")
        self.mlp_backward = mlp_backward
        self._vec_stream = self._vec_events = None
        self.pipeline_min_samples = 4 * 65536   # below this the pieces are too small to fill the chip
        self._scatter_stream = None
        self._piece_arenas = None
        self._piece_events = [torch.cuda.Event() for _ in range(4)] if dev.type == "cuda" else None
        self.scaler = ops.grad_scaler(dev, init_scale=grad_scale, growth_interval=scaler_growth_interval)
        # Table-gradient scatter: "binned" = radix partition + LDS accumulation (csrc/scatter.hip: no memory-side atomics;
        # level tables of up to 2^19 entries), "atomic" = the level-major atomic kernel, "auto" = binned when the model
        # fits it. The workspace holds the record queues of the largest batch a step can render (1.1 x samples_max).
        if table_scatter not in ("auto", "binned", "atomic"):
            raise ValueError("table_scatter must be 'auto', 'binned' or 'atomic'")
        fits = ops.ScatterWorkspace.supports(m.max_level_entries, m.num_segments)
        if table_scatter == "binned" and not fits:
            raise ValueError(f"table_scatter='binned' serves level tables of up to {ops.ScatterWorkspace.MAX_LEVEL_ENTRIES} "
                             f"entries; this model has {m.max_level_entries}")
        self.scatter_ws = None
        if dev.type == "cuda" and fits and table_scatter != "atomic":
            self.scatter_ws = ops.ScatterWorkspace(int(samples_max_batch_size * 1.1) + 1024, m.num_segments,
                                                   m.max_level_entries, dev)
        self.evaluated = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss_sums = torch.zeros(3, dtype=torch.float32, device=dev)
        # composite + loss + composite backward as one launch (hrf_render_loss_fused) instead of three: same bits, two launches and
        # two re-reads of the rays' samples less on the step's critical path (False: the three reference-shaped calls)
        self.fused_render_loss = True
        self.fused_encode_density = True   # hrf_encode4d_density_fwd instead of hrf_encode4d_fwd + hrf_density_mlp_fwd (same bits)
        m._refresh_half()
        self._table_ranges = []   # [start, end) of every segment's four tables inside table_params (elements)
        entries, t_off = [], 0
        vec_n = m.vectors[0].numel()
        p_t, g_t, ea_t, eas_t = m.table_params.data, self._grads[0], self.exp_avg[0], self.exp_avg_sq[0]
        self.shards = None        # TableShardExchange of the sharded exchange
        if self.exchange == "sharded":
            tr, o = [], 0
            for e in m.entries_per_segment:
                tr.append((o * 2, (o + 4 * e) * 2))
                o += 4 * e
            self.shards = TableShardExchange(tr, world_size, self.rank, process_group)
            if self._dp and dev.type == "cuda":
                self.shards.self_check(dev)          # raises when the backend's in-place collectives do not behave like RCCL's
        for sidx, e in enumerate(m.entries_per_segment):
            a, b = t_off * 2, (t_off + 4 * e) * 2
            self._table_ranges.append((a, b))
            if self.shards is not None:
                oa, ob = self.shards.own_ranges[sidx]
                entries.append((p_t[oa:ob], g_t[oa:ob], ea_t[oa:ob], eas_t[oa:ob], m._tables_h[oa:ob], 1 + sidx))
                for za, zb in ((a, oa), (ob, b)):          # the other ranks' shards: their local gradients are only zeroed
                    if zb > za:
                        entries.append((None, g_t[za:zb], None, None, None, 1 + sidx))
            else:
                entries.append((p_t[a:b], g_t[a:b], ea_t[a:b], eas_t[a:b], m._tables_h[a:b], 1 + sidx))
            va, vb = sidx * vec_n, (sidx + 1) * vec_n
            entries.append((m.vectors.data.view(-1)[va:vb], self._grads[1][va:vb], self.exp_avg[1].view(-1)[va:vb],
                            self.exp_avg_sq[1].view(-1)[va:vb], None, 1 + sidx))
            t_off += 4 * e
        entries.append((m.sigma_params.data, self._grads[2], self.exp_avg[2], self.exp_avg_sq[2], m._sigma_h, 0))
        entries.append((m.color_params.data, self._grads[3], self.exp_avg[3], self.exp_avg_sq[3], m._color_h, 0))
        if m.camera_embedding_dim > 0:
            entries.append((m.camera_embeddings.weight.data.view(-1), self._grads[4], self.exp_avg[4].view(-1),
                            self.exp_avg_sq[4].view(-1), None, 0))
        self._adam_desc = ops.adam_descriptors(entries, dev)
        self._adam_ws = ops.adam_workspace(dev) if dev.type == "cuda" else None
        self._adam_count, self._adam_total = len(entries), total
        # device-resident batch collection (one host sync per batch-growing iteration); needs the loader to expose
        # its HBM-resident pool tables the way SyntheticDataLoader does
        self.collector = None
        if fast_collect and hasattr(loader, "pixel_colors") and loader.pixel_colors.is_cuda:
            self.collector = StepCollector(model, loader, samples_max_batch_size, rays_initial_batch_size)
            self.pipeline_pieces = self._pipeline_pieces       # (the setter tells the collector)
            if self._dp and self.exchange == "sharded":
                # the next step's sampler stages are issued under the all-gather of the fp16 tables (train_step), when the
                # CUs have nothing else to do, instead of behind the march
                self.collector.auto_prefetch = False
        self._tables_pending = None          # finish() of the all-gather of the fp16 tables still in flight
        # time_exchange: record an event pair per step around the gradient exchange (table reduce-scatter / all-reduce issued ->
        # the compute stream has waited for it and for the small all-reduce; the vector-gradient kernel runs inside that window)
        self.time_exchange = False
        self.exchange_events = []
        self.collectives_used = set()
        # checkpoints: in the sharded exchange a rank's fp32 masters / moments are current only on its own shards, so
        # anything that serialises the model gathers first (collective: every rank must call it)
        # (weak references: the model must not keep the engine and its process group alive, and a second engine on the same model
        # simply replaces the hooks of the first)
        import weakref
        self._masters_fresh = True           # sharded exchange: every rank holds every shard's fp32 masters / moments (no step yet)
        model._master_sync = weakref.WeakMethod(self._model_master_sync)
        model._tables_ready = weakref.WeakMethod(self._finish_tables)

    # ------------------------------------------------------------------ pieces
    @property
    def _dp(self) -> bool:
        """The step runs its data-parallel form (gradient exchange before the optimizer)."""
        return self.world_size > 1 or getattr(self, "force_collectives", False)

    @property
    def pipeline_pieces(self) -> int:
        return self._pipeline_pieces

    @pipeline_pieces.setter
    def pipeline_pieces(self, n: int) -> None:
        """The pieces need ray-aligned cut points of a batch in DRAW order (StepCollector hands none over for a frame-ordered
        batch): asking for pieces turns the frame ordering off -- and with it the binned scatter, see _table_scatter --
        instead of silently doing nothing; 1 turns it back on."""
        self._pipeline_pieces = int(n)
        if getattr(self, "collector", None) is not None:
            if self._pipeline_pieces > 1 and self.collector.sort_batch:
                warnings.warn("TrainEngine.pipeline_pieces > 1 turns the frame-ordered batch layout off, and with it the binned "
                              "table-gradient scatter (the atomic scatter takes over): measured slower than one piece on MI355X "
                              "(DESIGN.md section 4, what did not pay)", stacklevel=2)
            self.collector.sort_batch = self._pipeline_pieces <= 1

    def lr(self) -> float:
        return self.lr0 * self.lr_decay ** min(self.sched_step / self.max_steps, 1.0)

    def collect_batch(self) -> (InputBatch, StepStats):
        """trainer.py:138-172."""
        st = StepStats()
        if self.collector is not None:
            with ops._span("phase_collect", 1):
                batch, st.num_rays_drawn, _ = self.collector.collect()   # sample statistics: collector.totals
            st.num_rays, st.num_samples = batch.num_rays, batch.num_samples
            return batch, st
        self.loader.batch_size = self.rays_initial
        total_rays = total_samples = 0
        batches = []
        while True:
            with ops._span("phase_sampler", 1):
                b = next(self.loader)
            st.num_samples_pre += b.num_samples
            with ops._span("phase_prune", 1):
                prune_samples(b, self.model, True)
            ev = getattr(b, "_num_evaluated", None)
            if ev is not None:  # samples the fused march actually encoded (device-side counter, no sync)
                self.evaluated += ev.sum()
                b._num_evaluated = None
            batches.append(b)
            total_rays += self.loader.batch_size
            total_samples += b.num_samples
            if total_samples < 0.9 * self.samples_max:
                avg = total_samples / total_rays
                assert avg > 0, "There is probably a problem with the predicted geometry."
                self.loader.batch_size = int((self.samples_max - total_samples) / avg)
            else:
                break
        st.num_rays_drawn = total_rays
        with ops._span("phase_merge", 1):
            batch = merge_input_batches(batches, max_num_samples=int(self.samples_max * 1.1))
        st.num_rays, st.num_samples = batch.num_rays, batch.num_samples
        return batch, st

    def _exchange_ranges(self):
        """Element ranges of the table gradients that can be non-zero on some rank: the tables of the segments whose
        frames sit in the image pools. Needs loaders whose pools hold the same FRAMES on every rank (shared frame
        schedule, per-rank cameras: SyntheticDataLoader(seed=shared, camera_seed=per rank)); otherwise everything."""
        if not self.exchange_touched_only or not getattr(self.loader, "frame_synchronous", False):
            return None
        m = self.model
        segs = sorted({int(m._f2s_host[f]) for f in self.loader.frames_superset()})
        ranges: List[List[int]] = []
        for sidx in segs:
            a, b = self._table_ranges[sidx]
            if ranges and ranges[-1][1] == a:
                ranges[-1][1] = b     # neighbouring segments: one message
            else:
                ranges.append([a, b])
        return [(a, b) for a, b in ranges]

    def _exchange_segments(self) -> List[int]:
        """Temporal segments whose table gradients can be non-zero on some rank (see _exchange_ranges); all when unknown."""
        m = self.model
        if not self.exchange_touched_only or not getattr(self.loader, "frame_synchronous", False):
            return list(range(m.num_segments))
        return sorted({int(m._f2s_host[f]) for f in self.loader.frames_superset()})

    def _exchange_groups(self, segs: Sequence[int]) -> List[List[int]]:
        """`segs` (ascending) cut into groups of CONSECUTIVE segment ids (hrf_scatter_accumulate takes a range of ids): a gap in
        the ids starts a new group; beyond that, up to `exchange_groups` groups in all, handed to the runs of neighbouring ids by
        bytes (the run with the most bytes per group gets the next one) and cut inside a run at equal shares of its bytes."""
        runs: List[List[int]] = []
        for sidx in segs:
            if runs and runs[-1][-1] + 1 == sidx:
                runs[-1].append(sidx)
            else:
                runs.append([sidx])
        size = lambda ids: sum(self._table_ranges[i][1] - self._table_ranges[i][0] for i in ids)
        parts = [1] * len(runs)
        for _ in range(max(self.exchange_groups - len(runs), 0)):
            cand = [i for i, r in enumerate(runs) if parts[i] < len(r)]
            if not cand:
                break
            parts[max(cand, key=lambda i: size(runs[i]) / parts[i])] += 1
        out: List[List[int]] = []
        for run, k in zip(runs, parts):
            total, acc, cur, done = size(run), 0, [], 0
            for pos, sidx in enumerate(run):
                cur.append(sidx)
                acc += size([sidx])
                left_ids, left_groups = len(run) - pos - 1, k - done - 1
                if left_groups > 0 and (acc * k >= total * (done + 1) or left_ids == left_groups):
                    out.append(cur)
                    cur, done = [], done + 1
            if cur:
                out.append(cur)
        return out

    def gather_master_tables(self) -> None:
        """Sharded exchange: bring the fp32 master tables (and Adam moments) of every segment up to date on every rank;
        between such calls a rank's masters are current only on its own shards. COLLECTIVE: every rank must call it (a call
        that finds the masters fresh -- nothing stepped since the last gather -- issues nothing). TrainEngine.state_dict() calls
        it; HumanRF.state_dict() / reference_state_dict() do NOT when more than one rank is involved: they raise on stale
        masters (see _model_master_sync), so a forgotten gather is an error message, never a deadlock or a stale checkpoint."""
        if self.exchange != "sharded" or self.shards is None or self._masters_fresh:
            return                      # (nothing was stepped since the last gather: no collective is issued)
        self._finish_tables()
        every = list(range(self.model.num_segments))
        for t in (self.model.table_params.data, self.exp_avg[0], self.exp_avg_sq[0]):
            self.shards.all_gather(t, every)
        self._masters_fresh = True

    @property
    def masters_fresh(self) -> bool:
        """True when this rank's fp32 master tables and Adam moments are complete (always, outside the sharded exchange)."""
        return self.exchange != "sharded" or self.shards is None or self._masters_fresh

    def _model_master_sync(self) -> None:
        """What HumanRF.state_dict() / reference_state_dict() call first. Serialising the model is something ONE rank usually
        does (`if rank == 0: torch.save(model.state_dict())`); a collective hidden in there would deadlock the job, so with more
        than one rank stale masters are an error here, with the remedy in the message. (One rank -- including the forced
        one-rank group of bench.py --force-collectives -- gathers on the spot: nobody else has to take part.)"""
        if self.masters_fresh:
            return
        if self.world_size > 1:
            raise RuntimeError("HumanRF.state_dict(): the fp32 master tables of the other ranks' shards are stale (sharded gradient "
                               "exchange). Call engine.gather_master_tables() on EVERY rank first -- after that any single rank may "
                               "serialise the model -- or checkpoint through engine.state_dict(), which is collective.")
        self.gather_master_tables()

    def _finish_tables(self) -> None:
        """Make the current stream wait for the all-gather of the fp16 tables issued behind the last optimizer launch (the
        model calls this before anything reads the tables: HumanRF._refresh_half)."""
        pending, self._tables_pending = self._tables_pending, None
        if pending is not None:
            pending()

    def state_dict(self) -> dict:
        """Everything a resume needs (the reference saves model, optimizer, scheduler and scaler state, trainer.py:528-560):
        the model in the reference's layout, Adam's moments and per-group step counts, the step counters and the GradScaler.
        Gathers the sharded masters and moments first (COLLECTIVE in the sharded exchange)."""
        self.gather_master_tables()
        torch.cuda.synchronize() if self.model.table_params.is_cuda else None
        return {"model": self.model.reference_state_dict(),
                "exp_avg": [t.detach().clone() for t in self.exp_avg],
                "exp_avg_sq": [t.detach().clone() for t in self.exp_avg_sq],
                "opt_state": self.opt_state.clone(), "scaler": self.scaler.clone(),
                "step": self.step, "sched_step": self.sched_step}

    def load_state_dict(self, sd: dict) -> None:
        self._finish_tables()
        self.model.load_reference_state_dict(sd["model"])
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)
        self.opt_state.copy_(sd["opt_state"])
        self.scaler.copy_(sd["scaler"])
        self.step, self.sched_step = int(sd["step"]), int(sd["sched_step"])
        self.model._refresh_half()

    def _table_scatter(self, xyzt, seg, enc, vectors, d_feats) -> None:
        """d_tables += the table half of Decomposition4D's backward (level-major dY from hrf_mlp_bwd)."""
        m = self.model
        ws = self.scatter_ws
        # The binned scatter wants tiles of ONE temporal segment: a batch laid out by frame (the collector's default) or a
        # one-segment model. On any other layout every sample of a tile's minority segments takes its direct path (~1000
        # atomics per sample against the level-major kernel's 58), so such batches go to the level-major kernel.
        if ws is not None and xyzt.shape[0] <= ws.samples and (self._batch_sorted or m.num_segments == 1):
            ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, self._grads[0], ws,
                                           flags=self.flags, grad_boundary=self._gb_tables)
        else:
            ops.encode4d_bwd(xyzt, seg, enc, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, self._grads[0], None,
                             level_major=True, grad_boundary=self._gb_tables, flags=self.flags)

    def _pieces(self, ib: InputBatch) -> List[tuple]:
        """(ray_lo, ray_hi, sample_lo, sample_hi) of the pieces the step is fed in. One piece = the whole batch; more when
        the collector handed over ray-aligned cut points and the step runs on one GPU (see train_step)."""
        R, N = ib.num_rays, ib.num_samples
        cuts = getattr(ib, "_cuts", None)
        n = self.pipeline_pieces
        if self._dp or n <= 1 or not cuts or N < self.pipeline_min_samples:
            return [(0, R, 0, N)]
        pts = [(0, 0)] + [cuts[k] for k in ((1,) if n == 2 else (0, 1, 2))] + [(R, N)]
        out = [(pts[i][0], pts[i + 1][0], pts[i][1], pts[i + 1][1]) for i in range(len(pts) - 1)]
        if any(rh <= rl or sh <= sl for rl, rh, sl, sh in out):
            return [(0, R, 0, N)]
        return out

    def train_step(self, ib: InputBatch) -> None:
        """trainer.py:229-255 with explicit kernels.

        Optional (pipeline_pieces > 1, one GPU): the batch is fed in ray-aligned pieces (halves / quarters): the loss is a
        mean over rays, every ray's samples stay in one piece, and all gradients are accumulated with atomics, so the
        result is the sum the single pass produces (tests/test_gpu_fullsize.py). The table / vector gradient scatter of
        piece k then runs on a second stream while the main stream computes forward + MLP backward of piece k+1. Measured:
        no gain (see __init__), so the default is one pass."""
        m = self.model
        dev = ib.ray_origins.device
        R = ib.num_rays
        self._finish_tables()      # (a caller that steps without collecting: nothing below goes through _refresh_half)
        self._batch_sorted = bool(getattr(ib, "_sorted_by_frame", False))
        S = self.internal_grad_scale   # x the device-side GradScaler's scale, applied inside the loss kernel
        gt = ib.rgba.contiguous()
        background = torch.rand(R, 3, dtype=torch.float32, device=dev)  # trainer.py:237
        t_all = ib.sample_distances.reshape(-1).contiguous()
        ray_idx_all = ib.ray_indices.contiguous()
        origins = ib.ray_origins.contiguous()
        dirs = ib.ray_directions.contiguous()
        cams = ib.camera_numbers.reshape(-1).contiguous()
        frames = ib.frame_numbers.reshape(-1).contiguous()
        vectors = m.vectors.detach()
        sw1, sw2 = m._sigma_w()
        cw1, cw2, cw3 = m._color_w()
        E = m.camera_embedding_dim
        emb = m.camera_embeddings.weight.detach() if E > 0 else None
        g = self._grads
        kin = m.color_in_pad
        pieces = self._pieces(ib)
        side = None
        exchanged = None     # segments whose tables went through the sharded exchange in this step
        arena_before = ops.ARENA
        if len(pieces) > 1:
            if self._scatter_stream is None:
                self._scatter_stream = torch.cuda.Stream(device=dev)
                self._piece_arenas = [ops.Arena() for _ in range(4)]
            side = self._scatter_stream
            side.wait_stream(torch.cuda.current_stream())   # gradient buffers zeroed by the previous optimizer launch
        try:
            for k, (rl, rh, sl, sh) in enumerate(pieces):
                if side is not None:
                    ops.ARENA = self._piece_arenas[k]       # the scatter of piece k reads its buffers while k+1 is computed
                t, ray_idx = t_all[sl:sh], ray_idx_all[sl:sh]
                Rk = rh - rl
                # ---- forward (per-sample kernels index the per-ray arrays with the batch-wide ray ids)
                xyzt, seg = ops.query_prep(origins, dirs, frames, ray_idx, t, None, m.frame_numbers_to_segment_numbers,
                                           m.frame_numbers_to_normalized_local_frame_numbers)
                if self.fused_encode_density and dev.type == "cuda":
                    # the render pass's encoding and sigma_net in one launch (feature rows handed over in LDS): same bits
                    feats, enc, h, sigma = ops.encode4d_density_fwd(xyzt, seg, m._tables_h, vectors, m._seg_meta, m.num_segments, sw1, sw2,
                                                                    float(m.density_scale))
                else:
                    feats, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, vectors, m._seg_meta, m.num_segments, save_enc=True)
                    h, sigma = ops.density_mlp_fwd(feats, sw1, sw2, float(m.density_scale))
                rgb = ops.color_mlp_fwd(dirs, ray_idx, h, emb, cams, E, E > 0, cw1, cw2, cw3, geo_dim=m.geometry_feature_dim)
                given = getattr(ib, "_ray_start", None) if len(pieces) == 1 else None
                ray_start = given if given is not None else ops.ray_offsets(ray_idx, rh)[rl:]   # offsets of the piece's rays inside the piece
                bg = background[rl:rh]
                # ---- composite, loss and the composite's backward; the loss also marks the temporal segments of the batch's
                # rays: the parameters that receive a gradient in the reference (humanrf.py:159-163), the ones Adam steps
                if self.fused_render_loss and dev.type == "cuda":
                    # one launch, one wavefront per ray (bit-identical to the three below)
                    d_sigma, d_rgb, _, _ = ops.render_loss_fused(sigma, rgb, t, ray_start, bg, gt[rl:rh], Rk, self.delta, self.bce_w, S,
                                                                 self.loss_sums, frames[rl:rh],
                                                                 m.frame_numbers_to_segment_numbers, self._touched,
                                                                 scaler=self.scaler, norm_rays=R)
                else:
                    color, acc = ops.composite_fwd(sigma, rgb, t, ray_start, bg, Rk)
                    d_color, d_acc = ops.loss_fwd_bwd(color, acc, gt[rl:rh], bg, self.delta, self.bce_w, S, self.loss_sums,
                                                      frames[rl:rh], m.frame_numbers_to_segment_numbers, self._touched,
                                                      scaler=self.scaler, norm_rays=R)
                    d_sigma, d_rgb = ops.composite_bwd(sigma, rgb, t, ray_start, bg, d_color, d_acc, Rk)
                if self.mlp_backward == "split":
                    # colour network first (d_h = its geometry-input gradient + the truncated_exp backward of d_sigma), then
                    # sigma_net: two kernels at two wavefronts per SIMD instead of one at one; h comes from the forward
                    d_h = ops.color_mlp_bwd(dirs, ray_idx, h, emb, cams, E, E > 0, cw1, cw2, cw3, d_rgb, *m.split_color(g[3]),
                                            g[4] if E > 0 else None,
                                            self.flags, d_sigma=d_sigma, density_scale=float(m.density_scale), arena=True,
                                            geo_dim=m.geometry_feature_dim)
                    d_feats = ops.density_mlp_bwd(feats, sw1, sw2, d_h, g[2][:2048], g[2][2048:], self.flags, level_major=True,
                                                  grad_boundary=self._gb_mlp)
                else:
                    d_feats = ops.mlp_bwd(feats, dirs, ray_idx, emb, cams, E, E > 0, sw1, sw2, cw1, cw2, cw3,
                                          float(m.density_scale), d_rgb, d_sigma, g[2][:2048], g[2][2048:], *m.split_color(g[3]),
                                          g[4] if E > 0 else None,
                                          self.flags, level_major=True, grad_boundary=self._gb_mlp, geo_dim=m.geometry_feature_dim)
                # ---- backward of the encoding (+ data-parallel gradient exchange)
                if side is not None:
                    ev = self._piece_events[k]
                    ev.record()
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        self._table_scatter(xyzt, seg, enc, vectors, d_feats)
                        ops.encode4d_bwd(xyzt, seg, enc, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, None, g[1], level_major=True)
                elif not self._dp:           # table scatter and vector scatter, timed separately
                    binned = (self.scatter_ws is not None and xyzt.shape[0] <= self.scatter_ws.samples
                              and (self._batch_sorted or m.num_segments == 1))
                    # (only under the BINNED table scatter: the level-major one is bound by the same atomic unit as the vector
                    # half -- measured with one 2^18 segment: vectors 0.62 ms under it against 0.16 behind it)
                    if self.overlap_vector_scatter and binned and dev.type == "cuda":
                        # the vector half (bound by memory-side atomic requests, next to no LDS) on a second stream under the
                        # table half (emit: VALU / LDS-slot bound, three workgroups per CU since round 5; accumulate: one 128 KB
                        # workgroup per CU): both read d_feats, they write different gradient buffers. What it hides today is
                        # small -- 0.03 ms under the accumulate kernel, nothing under the emit kernel
                        # (profiles/r05_scatter_variants.txt) -- and it costs nothing
                        if self._vec_stream is None:
                            self._vec_stream = torch.cuda.Stream(device=dev)
                            self._vec_events = (torch.cuda.Event(), torch.cuda.Event())
                        e0, e1 = self._vec_events
                        if self.vector_scatter_under == "accumulate":
                            # emit first, alone; the vector kernel starts with the accumulate kernel (one 128 KB workgroup of 71-register
                            # wavefronts per CU: room for the vector kernel's 68-register wavefronts next to it, which the emit
                            # kernel's 18 wavefronts per CU do not leave)
                            ops.scatter_emit(xyzt, seg, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, self._grads[0], self.scatter_ws,
                                             grad_boundary=self._gb_tables)
                        e0.record()
                        with torch.cuda.stream(self._vec_stream):
                            self._vec_stream.wait_event(e0)
                            ops.encode4d_bwd(xyzt, seg, enc, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, None, g[1],
                                             level_major=True)
                            e1.record()
                        if self.vector_scatter_under == "accumulate":
                            ops.scatter_accumulate(m._seg_meta, m.num_segments, self._grads[0], self.scatter_ws, flags=self.flags)
                        else:
                            self._table_scatter(xyzt, seg, enc, vectors, d_feats)
                        torch.cuda.current_stream().wait_event(e1)
                    else:
                        self._table_scatter(xyzt, seg, enc, vectors, d_feats)
                        ops.encode4d_bwd(xyzt, seg, enc, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, None, g[1], level_major=True)
                else:
                    # Table gradients first, and their exchange in PIECES ("overlap with the remaining backward"):
                    # the batch is laid out by frame and the scatter's tiles never straddle a temporal segment, so once the record
                    # queues exist (emit) the segments are accumulated group by group, and a group's reduce-scatter (or
                    # all-reduce) is issued the moment its accumulate launch is enqueued -- RCCL orders a collective behind what
                    # the compute stream holds when it is called, so group k travels over the links while groups k+1.. are
                    # accumulated and the vector half of the backward runs. (Rounds 3-5 issued everything after the whole scatter.)
                    ws = self.scatter_ws
                    timing = self.time_exchange and dev.type == "cuda"
                    ev_x = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timing else None
                    segs_x = self._exchange_segments()
                    # The SEQUENCE of collectives must be the same on every rank: the groups follow from the exchanged segments and
                    # the model alone. Whether THIS rank's batch can go through the binned scatter (it is laid out by frame and fits
                    # the workspace: always, with the collector) only decides how its gradients get there -- a rank that cannot
                    # scatters everything up front and issues the same group collectives.
                    groups = self._exchange_groups(segs_x) if (ws is not None and self.exchange_groups > 1) else [list(segs_x)]
                    binned = (ws is not None and xyzt.shape[0] <= ws.samples and (self._batch_sorted or m.num_segments == 1)
                              and len(groups) > 1)
                    pendings = []
                    if self.shards is not None:
                        self.shards.bytes_issued, self.shards.issue_log = 0, []
                    self.exchange_bytes, self.exchange_issue_log = 0, []
                    if binned:
                        ops.scatter_emit(xyzt, seg, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, self._grads[0], ws,
                                         grad_boundary=self._gb_tables)
                    else:
                        self._table_scatter(xyzt, seg, enc, vectors, d_feats)
                    if ev_x is not None:
                        ev_x[0].record()
                    signalled = (binned and self.exchange_signalled and dev.type == "cuda" and len(groups) <= 8
                                 and ops.can_stream_wait_value())
                    main_stream = torch.cuda.current_stream() if dev.type == "cuda" else None
                    if signalled:
                        if self._group_done is None:
                            self._group_done = torch.zeros(8, dtype=torch.int64, device=dev)
                            self._sig_stream = torch.cuda.Stream(device=dev)
                            self._sig_stream.wait_stream(main_stream)            # (the counters are zero before anything waits on them)
                        per_seg = ops.scatter_signals_per_segment(ws)
                        goal = [self._group_goal[gi] + per_seg * (grp[-1] - grp[0] + 1) for gi, grp in enumerate(groups)]
                        ops.scatter_accumulate_signalled(m._seg_meta, m.num_segments, self._grads[0], ws, self.flags, groups, self._group_done)
                        # (committed only once the launch is enqueued: a goal the counters never reach would park its waiter for good)
                        self._group_goal[:len(groups)] = goal
                        self.exchange_issue_log_mode = "one accumulate launch, signalled per group"
                    else:
                        self.exchange_issue_log_mode = "one accumulate launch per group" if binned else "scatter first"
                    for gi, grp in enumerate(groups):
                        if binned and not signalled:
                            ops.scatter_accumulate(m._seg_meta, m.num_segments, self._grads[0], ws, flags=self.flags,
                                                   seg_first=grp[0], seg_count=grp[-1] - grp[0] + 1)
                        if signalled:
                            # the collective is handed to the backend from a stream that holds nothing but the wait for this group's
                            # count: the backend orders it behind that stream, i.e. behind the moment the group's gradients are complete
                            with torch.cuda.stream(self._sig_stream):
                                ops.stream_wait_value64(self._group_done, gi, self._group_goal[gi])
                                if self.exchange == "sharded":
                                    pendings.append(self.shards.reduce_scatter(g[0], grp))
                                else:
                                    rng = [(self._table_ranges[grp[0]][0], self._table_ranges[grp[-1]][1])]
                                    pendings.append(allreduce_gradients(self.flat_grad, self._big, self.world_size, self.group,
                                                                        self.transport_dtype, wire=self._wire, average=False, tail=False,
                                                                        wait=False, head_ranges=rng, force=self.force_collectives))
                            if self.exchange != "sharded":
                                self.collectives_used.add("all_reduce (table gradients)")
                                self.exchange_bytes += sum(b - a for a, b in rng) * (4 if self.transport_dtype in (None, torch.float32) else 2)
                                self.exchange_issue_log.append(("all_reduce", tuple(grp)))
                            continue
                        if self.exchange == "sharded":
                            pendings.append(self.shards.reduce_scatter(g[0], grp))
                        else:
                            rng = self._exchange_ranges() if len(groups) == 1 else [(self._table_ranges[grp[0]][0], self._table_ranges[grp[-1]][1])]
                            pendings.append(allreduce_gradients(self.flat_grad, self._big, self.world_size, self.group,
                                                                self.transport_dtype, wire=self._wire, average=False, tail=False,
                                                                wait=False, head_ranges=rng, force=self.force_collectives))
                            self.collectives_used.add("all_reduce (table gradients)")
                            self.exchange_bytes += sum(b - a for a, b in (rng if rng is not None else [(0, self._big)])) * \
                                (4 if self.transport_dtype in (None, torch.float32) else 2)
                            self.exchange_issue_log.append(("all_reduce", tuple(grp)))
                    if self.exchange == "sharded":
                        exchanged = segs_x
                        self.exchange_bytes, self.exchange_issue_log = self.shards.bytes_issued, list(self.shards.issue_log)
                    ops.encode4d_bwd(xyzt, seg, enc, vectors, m._seg_meta, m.num_segments, d_feats, 1.0, None, g[1], level_major=True)
                    # found_inf and the touched flags ride behind the small gradients (sum over ranks = logical OR)
                    self._flag_f[0:1].copy_(self.flags)
                    self._flag_f[1:].copy_(self._touched)
                    allreduce_gradients(self.flat_grad, self._big, self.world_size, self.group, self.transport_dtype,
                                        wire=self._wire, average=False, head=False, force=self.force_collectives)
                    self.collectives_used.add("all_reduce (vectors, MLPs, embeddings, flags)")
                    if ev_x is not None:        # the compute stream has nothing left to do but wait: what follows is EXPOSED
                        ev_x[1].record()
                    for pending in pendings:
                        pending()
                    if ev_x is not None:        # first table collective issued -> the compute stream has waited for all of them
                        ev_x[2].record()
                        self.exchange_events.append(ev_x)
                    self.flags.copy_(self._flag_f[0:1] > 0)
                    self._touched.copy_(self._flag_f[1:] > 0)
                    self._flag_f.zero_()
        finally:
            ops.ARENA = arena_before
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        # ---- optimizer (GradScaler.step semantics: skipped on found_inf; Adam's per-parameter step counts and the
        # bookkeeping live on the device) + LR schedule
        self.step += 1
        # (data parallel: the gradients are SUMS over the ranks; the mean's 1/N goes into the optimizer's unscale factor)
        ops.adam_multi(self._adam_desc, self._adam_count, self.num_groups, self._adam_total, self.lr(), self.betas[0],
                       self.betas[1], self.eps, S * self.world_size, self.opt_state, self._adam_ws, scaler=self.scaler)
        m.mark_half_fresh()
        if exchanged is not None:
            # every rank's freshly cast fp16 shard -> all ranks, in place in the table copy the kernels gather from (the
            # fp32 masters of the other ranks' shards stay behind; gather_master_tables() refreshes them for checkpoints).
            # One collective per segment, issued now and NOT waited for: the next reader of the tables waits
            # (HumanRF._refresh_half -> _finish_tables: the next step's prune march), and the next step's sampler stages,
            # which do not read the tables, are issued under the exchange.
            self._tables_pending = self.shards.all_gather(m._tables_h, exchanged, wait=False)
            self._masters_fresh = False      # the other ranks' shards of the fp32 masters / moments stayed behind
            self.collectives_used |= self.shards.collectives_used
            if self.collector is not None and not self.collector.auto_prefetch:
                self.collector.prefetch()
        self.sched_step += 1

    def exchange_ms(self, clear: bool = True):
        """(steps, mean ms issued, mean ms exposed) of the gradient-exchange windows recorded since the last call (time_exchange =
        True); one sync. issued: the first table collective is handed to the backend -> the compute stream has waited for all of
        them (and for the small all-reduce); the remaining accumulate launches and the vector half of the backward run inside
        that window. exposed: the part of it in which the compute stream had nothing left to run and only waited."""
        evs = self.exchange_events
        if clear:
            self.exchange_events = []
        if not evs:
            return 0, None, None
        torch.cuda.synchronize()
        issued = [e[0].elapsed_time(e[2]) for e in evs]
        exposed = [e[1].elapsed_time(e[2]) for e in evs]
        return len(issued), sum(issued) / len(issued), sum(exposed) / len(exposed)

    def found_inf(self) -> int:
        """Host check (one sync): number of steps skipped because a 16-bit gradient overflowed since the last call.
        (The scale's backoff already happened on the device, in the optimizer launch of the skipped step.)"""
        st = self.opt_state[:2].cpu()
        total = int(st[1]) + int(st[0] != 0)
        n = total - self._skipped_seen
        self._skipped_seen = total
        return n

    @property
    def grad_scale(self) -> float:
        """The GradScaler's current scale (one sync)."""
        return ops.grad_scaler_state(self.scaler)["scale"]

    def optimizer_steps(self) -> List[int]:
        """Adam's step count per optimizer group (0: MLPs / embeddings, 1 + s: segment s); one sync."""
        return self.opt_state[4:4 + self.num_groups].cpu().tolist()

    def replace_next(self) -> None:
        """One synchronous pool-replacement step of the loader (loaders without the background thread). The loader
        orders the slot write after the sampler launches that read the pool (events recorded by `pool_reader`)."""
        self.loader.replace_next()

    def train_iteration(self) -> StepStats:
        if hasattr(self.loader, "tick"):
            self.loader.tick()   # the background replacer may refill pool slots while this step runs
        batch, st = self.collect_batch()
        self.loss_sums.zero_()
        with ops._span("phase_train_step", 1):
            ops.ARENA = self._arena  # step-persistent output buffers: nothing below outlives the step
            try:
                self.train_step(batch)
            finally:
                ops.ARENA = None
        st.sums = self.loss_sums
        return st

    @staticmethod
    def psnr_from_sums(sums: torch.Tensor, num_rays: int) -> float:
        mse = float(sums[2].item()) / (3.0 * max(num_rays, 1))
        return -10.0 * math.log10(max(mse, 1e-20))  # trainer.py:218-223
