"""Row-sharded search across the GPUs of one node.

The reference is single-process; its only partitioning is `scan_parallel`'s contiguous row chunks
merged by `merge_partial_heaps` (crates/frankensearch-index/src/search.rs:1013-1036,1704-1720).
The same shape across GPUs: rank r owns the contiguous rows [r*ceil(N/W), ...), reports GLOBAL row
ids, so the (score, row) tie-break is shard-invariant; queries are replicated; each rank produces
[B, k] packed hits (8 bytes each); ONE all-gather (RCCL over xGMI when the backend is nccl) of
B*k*8 bytes per rank; every rank merges the W lists with the reference order.  No all-reduce, no row
exchange.

Hybrid layout (round 5): W = G x S ranks as G query groups x S row shards — rank r scans row shard r % S (the caller builds its
index over shard_range(N, r % S, S)) for the queries of group r // S, a contiguous ceil(B / G) of the batch; ONE all-gather of
equal-sized lists over all W ranks, one merge per group over its S lists.  A shard's step has a fixed part that does not shrink
with its rows, so fewer, larger row shards x several query groups can beat W row shards (every GPU holds the slab many times over).

The compute backend is injected: `GpuShardBackend` (libfsgpu.so) in production; the CPU test suite
injects an oracle-based stand-in to exercise the partitioning / collective / layout logic under gloo.
"""
from __future__ import annotations

from typing import Optional, Protocol, Tuple

import torch
import torch.distributed as dist

EMPTY = -1  # 0xFFFFFFFFFFFFFFFF as int64: padding entry of a packed list


def shard_range(nrows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ceil-split: rank r owns [r*ceil(N/W), min(N, (r+1)*ceil(N/W)))."""
    per = (nrows + world - 1) // world
    lo = min(nrows, rank * per)
    hi = min(nrows, lo + per)
    return lo, hi


class ShardBackend(Protocol):
    def search_packed(self, queries: torch.Tensor, k: int) -> torch.Tensor:
        """[B, dim] f32 queries -> [B, k] int64 packed hits of THIS shard (global rows), best first."""

    def merge(self, gathered: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """[W, B, k] int64 packed -> (rows [B,k] int32-as-u32, scores [B,k] f32, counts [B] int32)."""


class GpuShardBackend:
    """libfsgpu.so on the current torch device / current stream."""
    supports_after_enqueue = True   # search_packed(after_enqueue=...): see ShardedVectorIndex.search_steps

    def __init__(self, index, device: torch.device, batched: bool = False):
        self.index = index
        self.device = device
        self.batched = batched        # serve through the matrix-core batched path (64 queries per HBM pass)
        self.last_fallbacks = 0
        self.hook_fired = False
        self._cb = self._cb_ptr = self._hook = self._hook_error = None

    def search_packed(self, queries: torch.Tensor, k: int, after_enqueue=None) -> torch.Tensor:
        """after_enqueue: called (on this thread) once the scan's kernels are enqueued and before the call blocks on its stream
        (fsgpu_index_set_after_enqueue_hook) — batched path only; `self.hook_fired` tells whether it ran."""
        import ctypes as C
        from . import _lib
        from .errors import check

        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        b, dim = queries.shape
        out = torch.empty((b, k), dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.hook_fired = False
        if self.batched:
            fb = C.c_uint32()
            if after_enqueue is not None:
                if self._cb is None:   # one C callback object per backend; what it runs is whatever the call in flight set
                    def _run(_ctx):
                        self.hook_fired = True
                        try:
                            self._hook()
                        except BaseException as e:   # must not unwind through the C frame
                            self._hook_error = e
                    self._cb = C.CFUNCTYPE(None, C.c_void_p)(_run)
                    self._cb_ptr = C.cast(self._cb, C.c_void_p)
                self._hook, self._hook_error = after_enqueue, None
                check(_lib.lib().fsgpu_index_set_after_enqueue_hook(self.index._h, self._cb_ptr, None))
            try:
                check(_lib.lib().fsgpu_search_topk_batched_packed_device(self.index._h, queries.data_ptr(), b, dim, k, None,
                                                                         out.data_ptr(), stream, C.byref(fb)))
            finally:
                if after_enqueue is not None and not self.hook_fired:   # (the library clears a hook it has called)
                    _lib.lib().fsgpu_index_set_after_enqueue_hook(self.index._h, None, None)
            if after_enqueue is not None and self._hook_error is not None:
                raise self._hook_error
            self.last_fallbacks = fb.value
        else:
            check(_lib.lib().fsgpu_search_topk_packed_device(self.index._h, queries.data_ptr(), b, dim, k, None,
                                                             out.data_ptr(), stream))
        return out

    def search_unsharded(self, queries: torch.Tensor, k: int):
        """world == 1: the shard-local merge already is the final answer (no exchange, no second merge)."""
        from . import _lib
        from .errors import check

        if self.batched:
            return self.search_batched(queries, k)
        b, dim = queries.shape
        rows = torch.empty((b, k), dtype=torch.int32, device=self.device)
        scores = torch.empty((b, k), dtype=torch.float32, device=self.device)
        counts = torch.empty((b,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.lib().fsgpu_search_topk_device(self.index._h, queries.data_ptr(), b, dim, k, None, rows.data_ptr(),
                                                  scores.data_ptr(), counts.data_ptr(), stream))
        return rows, scores, counts

    def search_batched(self, queries: torch.Tensor, k: int):
        """Throughput path: 64 queries per HBM pass on the matrix cores, exact results (synchronises the stream)."""
        import ctypes as C
        from . import _lib
        from .errors import check

        b, dim = queries.shape
        rows = torch.empty((b, k), dtype=torch.int32, device=self.device)
        scores = torch.empty((b, k), dtype=torch.float32, device=self.device)
        counts = torch.empty((b,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fb = C.c_uint32()
        check(_lib.lib().fsgpu_search_topk_batched_device(self.index._h, queries.data_ptr(), b, dim, k, None,
                                                          rows.data_ptr(), scores.data_ptr(), counts.data_ptr(), stream,
                                                          C.byref(fb)))
        self.last_fallbacks = fb.value
        return rows, scores, counts

    # ---- the batched search in two halves (fsgpu_search_topk_batched_device_begin / _end) -----------------------------------
    @property
    def supports_pipelined_scans(self) -> bool:
        return bool(self.batched)

    def scan_begin(self, queries: torch.Tensor, k: int, packed: bool):
        """Enqueue the whole batched search of `queries` on the current stream and return (outputs, ticket) WITHOUT waiting:
        outputs = the packed [B, k] list (packed=True: a shard's half of a sharded search) or (rows, scores, counts).  The
        queries and the outputs must stay alive until scan_end(ticket)."""
        import ctypes as C
        from . import _lib
        from .errors import check

        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        b, dim = queries.shape
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ticket = C.c_int32(-1)
        if packed:
            out = torch.empty((b, k), dtype=torch.int64, device=self.device)
            check(_lib.lib().fsgpu_search_topk_batched_device_begin(self.index._h, queries.data_ptr(), b, dim, k, None, None, None, None,
                                                                    out.data_ptr(), stream, C.byref(ticket)))
        else:
            rows = torch.empty((b, k), dtype=torch.int32, device=self.device)
            scores = torch.empty((b, k), dtype=torch.float32, device=self.device)
            counts = torch.empty((b,), dtype=torch.int32, device=self.device)
            check(_lib.lib().fsgpu_search_topk_batched_device_begin(self.index._h, queries.data_ptr(), b, dim, k, None, rows.data_ptr(),
                                                                    scores.data_ptr(), counts.data_ptr(), None, stream, C.byref(ticket)))
            out = (rows, scores, counts)
        return out, (ticket.value, queries)   # (the ticket keeps the queries alive)

    def scan_end(self, ticket) -> int:
        """Wait for that search alone, read its verdicts, run its fallbacks; returns (and records) the number of fallbacks."""
        import ctypes as C
        from . import _lib
        from .errors import check

        fb, late = C.c_uint32(), C.c_uint32()
        check(_lib.lib().fsgpu_search_topk_batched_device_end_late(self.index._h, ticket[0], C.byref(fb), C.byref(late)))
        self.last_fallbacks = fb.value
        # queries answered in the end half (exact fallbacks + re-filtered on the f16 slab): their hits were enqueued just now, so whatever
        # was ordered behind begin's kernels (the exchange) has to be ordered behind these too
        self.last_late_answers = late.value
        return fb.value

    def merge(self, gathered: torch.Tensor, k: int):
        from . import _lib
        from .errors import check

        w, b, kk = gathered.shape
        rows = torch.empty((b, k), dtype=torch.int32, device=self.device)
        scores = torch.empty((b, k), dtype=torch.float32, device=self.device)
        counts = torch.empty((b,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.lib().fsgpu_merge_topk_device(self.device.index or 0, gathered.data_ptr(), b, w, kk, kk, b * kk, k,
                                                 rows.data_ptr(), scores.data_ptr(), counts.data_ptr(), stream))
        return rows, scores, counts


class ShardedVectorIndex:
    """One process per GPU; `search` is collective: every rank calls it with the same queries.

    `search_begin` / `search_end` split a search into the shard-local scan and the exchange + merge.  With
    `overlap=True` (CUDA tensors) the second half is enqueued on a side stream without blocking the host, so a caller that
    begins step i+1 right after ending step i runs the all-gather and the merge of step i underneath the scan of step
    i+1 (the batched scan synchronises only its own stream); `search_end` then returns tensors that are complete once the
    returned event has fired."""

    def __init__(self, backend: ShardBackend, group: Optional[dist.ProcessGroup] = None, overlap: bool = False,
                 force_collective: bool = False, query_groups: int = 1):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if query_groups < 1 or self.world % query_groups:
            raise ValueError("query_groups must divide the world size (query groups x row shards)")
        self.query_groups = query_groups
        self.row_shards = self.world // query_groups
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.my_group = self.rank // self.row_shards
        self.overlap = overlap
        # a one-rank group normally skips the all-gather; the single-GPU rehearsal of the N-rank path turns it on so that the
        # collective (RCCL), the side stream and the merge of the gathered layout all run
        self.force_collective = force_collective and dist.is_initialized()
        self._side = None
        self._scan_event = None   # recorded on the scan's stream right after the last search_begin returned (CUDA only)

    # ---- query groups: this rank's slice of a batch, and the padding that keeps every rank's list the same size -------------------
    def _per(self, b: int) -> int:
        return (b + self.query_groups - 1) // self.query_groups

    def _my_queries(self, queries: torch.Tensor) -> torch.Tensor:
        if self.query_groups == 1:
            return queries
        per = self._per(queries.shape[0])
        return queries[self.my_group * per:(self.my_group + 1) * per].contiguous()

    def _padded(self, local: Optional[torch.Tensor], b: int, k: int, like: torch.Tensor) -> torch.Tensor:
        """[nqg, k] -> [per, k] (EMPTY lists for the queries this group does not hold: the last groups of a ragged batch)."""
        per = self._per(b)
        if local is not None and local.shape[0] == per:
            return local
        out = torch.full((per, k), EMPTY, dtype=torch.int64, device=like.device)
        if local is not None and local.shape[0]:
            out[:local.shape[0]] = local
        return out

    def _merge(self, gathered: torch.Tensor, b: int, k: int):
        """[W, per, k] -> (rows, scores, counts) of the b queries: one merge per query group over its row shards' lists."""
        if self.query_groups == 1:
            return self.backend.merge(gathered, k)
        per = gathered.shape[1]
        parts = [self.backend.merge(gathered[g * self.row_shards:(g + 1) * self.row_shards].contiguous(), k)
                 for g in range(self.query_groups) if g * per < b]
        return tuple(torch.cat([p[i] for p in parts])[:b] for i in range(3))

    def search_begin(self, queries: torch.Tensor, k: int, after_enqueue=None) -> torch.Tensor:
        """[B, dim] -> this rank's packed [ceil(B / G), k] list (global rows, best first) for its query group."""
        b = queries.shape[0]
        mine = self._my_queries(queries)
        if mine.shape[0] == 0:
            local = None   # a ragged batch left this group without queries (the caller's hook, if any, runs in search_steps)
        elif after_enqueue is None or not getattr(self.backend, "supports_after_enqueue", False):
            local = self.backend.search_packed(mine, k)   # (the caller sees that the hook did not run and does its work itself)
        else:
            local = self.backend.search_packed(mine, k, after_enqueue=after_enqueue)
        local = self._padded(local, b, k, queries)
        self._batch = b
        # The batched scan synchronises its stream before it decides on fallbacks, but the fallback work itself (exact kernels, the
        # nested f16 re-filter, the scatter of their hits into `local`) is only ENQUEUED when the call returns.  This event covers
        # that tail — and nothing of the next scan, whose kernels are enqueued after it.
        self._scan_event = None
        if self.overlap and local.is_cuda:
            self._scan_event = torch.cuda.Event()
            self._scan_event.record(torch.cuda.current_stream(local.device))
        return local

    def search_steps(self, batch_of, first: int, n: int, k: int, after_scan=None, keep_all: bool = False):
        """n whole searches (queries batch_of(i), i = first .. first + n - 1) as a pipeline: the all-gather + merge of step i - 1
        are enqueued on the side stream from inside the scan call of step i, in the window after its kernels are enqueued and
        before it blocks on its stream — the exchange's GPU work AND its host work run under the scan.  Returns the last step's
        (rows, scores, counts) — or every step's with keep_all — complete when this returns."""
        if getattr(self.backend, "supports_pipelined_scans", False):
            return self._search_steps_pipelined(batch_of, first, n, k, after_scan, keep_all)
        outs, pending, prev, prev_event = [], None, None, None
        state = {"pending": None}

        def wait(p):
            if len(p) == 4:
                p[3].synchronize()
            return p[:3]

        for i in range(first, first + n):
            if pending is not None:
                done = wait(pending)              # the exchange enqueued a whole scan ago
                if keep_all:
                    outs.append(done)
            hook = None
            if prev is not None:
                def hook(prev=prev, prev_event=prev_event):
                    state["pending"] = self.search_end(prev, k, scan_event=prev_event)
            state["pending"] = None
            local = self.search_begin(batch_of(i), k, after_enqueue=hook)
            local_event = self._scan_event
            if after_scan is not None:
                after_scan()
            if prev is not None:
                # (a search that left the matrix-core path returns without calling the hook)
                pending = state["pending"] if state["pending"] is not None else self.search_end(prev, k, scan_event=prev_event)
            prev, prev_event = local, local_event
        if pending is not None:
            done = wait(pending)
            if keep_all:
                outs.append(done)
        last = None
        if prev is not None:
            last = wait(self.search_end(prev, k, scan_event=prev_event))
            outs.append(last)
        return outs if keep_all else last

    def _search_steps_pipelined(self, batch_of, first: int, n: int, k: int, after_scan=None, keep_all: bool = False):
        """search_steps over a backend whose scan comes in two halves (GpuShardBackend.scan_begin / scan_end): the scan of step
        i + 1 is ENQUEUED before the host waits for the scan of step i, so the GPU never idles between two scans (the blocking form
        pays the wake-up, the interpreter and the next call's first launch there: ~50 us of a 1.25M-row shard's 0.5 ms step); the
        all-gather + merge of step i go to the side stream behind an event recorded right after step i's kernels were enqueued —
        re-recorded behind its fallback work in the rare step that has any."""
        be = self.backend
        outs, exch = [], None          # exch: (rows, scores, counts, done event) of the step before the previous one
        prev = None                    # (local packed list, ticket, event) of the step whose scan is enqueued but not ended

        def finish(p):
            local, ticket, ev = p
            fb = be.scan_end(ticket) if ticket is not None else 0
            late = max(fb, getattr(be, "last_late_answers", 0)) if ticket is not None else 0
            if after_scan is not None:
                after_scan()
            if late and local.is_cuda:   # hits written by work that went to the stream only now (fallbacks, re-filtered queries): the exchange must wait for that too
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(local.device))
            return self.search_end(local, k, scan_event=ev)

        def wait(x):
            if len(x) == 4:
                x[3].synchronize()
            return x[:3]

        for i in range(first, first + n):
            full = batch_of(i)
            mine = self._my_queries(full)
            if mine.shape[0]:
                local, ticket = be.scan_begin(mine, k, packed=True)
            else:
                local, ticket = None, None
            local = self._padded(local, full.shape[0], k, full)
            self._batch = full.shape[0]
            ev = None
            if local.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(local.device))
            if prev is not None:
                nxt = finish(prev)     # ends step i - 1 (the GPU is already on step i), enqueues its exchange
                if exch is not None:
                    done = wait(exch)
                    if keep_all:
                        outs.append(done)
                exch = nxt
            prev = (local, ticket, ev)
        last = None
        if prev is not None:
            nxt = finish(prev)
            if exch is not None:
                done = wait(exch)
                if keep_all:
                    outs.append(done)
            last = wait(nxt)
            outs.append(last)
        elif exch is not None:
            last = wait(exch)
        return outs if keep_all else last

    def _gather(self, local: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.force_collective:
            return local.unsqueeze(0)
        # dim-0 concatenation form (accepted by both RCCL and gloo), viewed as [W, B, k]
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal path (no RCCL): stage through the host
            host = torch.empty((self.world * local.shape[0], local.shape[1]), dtype=local.dtype)
            dist.all_gather_into_tensor(host, local.cpu(), group=self.group)
            flat = host.to(local.device)
        else:
            flat = torch.empty((self.world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(flat, local, group=self.group)
        return flat.view(self.world, local.shape[0], local.shape[1])

    def search_end(self, local: torch.Tensor, k: int, scan_event=None):
        """All-gather of the W packed lists + merge.  Returns (rows, scores, counts[, event when overlap is on]).
        scan_event: recorded on the scan's stream when the call that produced `local` returned (search_begin): the side stream
        waits for exactly that — the scan AND whatever fallback work it left enqueued — instead of for the whole current stream,
        which by now may hold the NEXT scan's kernels."""
        b = getattr(self, "_batch", None) or local.shape[0]
        if not (self.overlap and local.is_cuda):
            return self._merge(self._gather(local), b, k)
        if self._side is None:
            self._side = torch.cuda.Stream(device=local.device)
        side = self._side
        if scan_event is not None:
            side.wait_event(scan_event)
        else:
            side.wait_stream(torch.cuda.current_stream(local.device))   # the scan that produced `local`
        local.record_stream(side)
        with torch.cuda.stream(side):
            out = self._merge(self._gather(local), b, k)
            done = torch.cuda.Event()
            done.record(side)
        return out + (done,)

    def search(self, queries: torch.Tensor, k: int):
        if self.world == 1 and not self.force_collective and hasattr(self.backend, "search_unsharded"):
            return self.backend.search_unsharded(queries, k)
        local = self.search_begin(queries, k)  # [B, k]
        out = self.search_end(local, k)
        if len(out) == 4:
            out[3].synchronize()
            out = out[:3]
        return out
