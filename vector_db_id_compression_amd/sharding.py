"""Inverted lists sharded over the GPUs of one node.

Every list is an independent codec unit: encode and decode need no exchange, so each rank owns a subset of
the lists (balanced by total length, not by count: list sizes are Zipf-like) and compresses / decompresses it
with its own GPU.  The only communication is the variable-size gather of decoded ids to the rank that runs a
search (custom_invlists_impl.cpp:477-525 decodes the lists a search touched): one point-to-point message per owning
rank over its direct xGMI link (RCCL send/recv posted as one batch; `gloo` in the CPU tests).  The
list -> (rank, local number) map is replicated (nlist * 12 bytes), so message sizes are known without a size exchange.

Nothing here loops over lists in Python except the LPT assignment (a heap pop per list): shards are cut with index
arithmetic and the gathered lists land in request order with one indexed copy per owning rank.
"""
import heapq

import numpy as np


def lpt_partition(sizes, world):
    """Longest-processing-time greedy: lists sorted by length, each to the currently lightest rank.
    -> owner int32[nlist]"""
    sizes = np.asarray(sizes, dtype=np.int64)
    owner = np.zeros(sizes.size, dtype=np.int32)
    if world <= 1 or sizes.size == 0:
        return owner
    heap = [(0, r) for r in range(world)]
    order = np.argsort(-sizes, kind="stable")
    sz = sizes[order].tolist()
    own = [0] * len(sz)
    for i, s in enumerate(sz):
        load, r = heap[0]
        own[i] = r
        heapq.heapreplace(heap, (load + s, r))
    owner[order] = np.asarray(own, dtype=np.int32)
    return owner


def _segment_index(starts, sizes, xp):
    """Concatenated index ranges [starts[k], starts[k] + sizes[k]) (numpy or torch)."""
    total = int(sizes.sum())
    if xp is np:
        cum = np.concatenate([[0], np.cumsum(sizes)[:-1]]) if sizes.size else np.zeros(0, np.int64)
        return np.repeat(starts - cum, sizes) + np.arange(total, dtype=np.int64)
    import torch

    cum = torch.cumsum(sizes, 0) - sizes
    return torch.repeat_interleave(starts - cum, sizes) + torch.arange(total, dtype=torch.int64, device=sizes.device)


class ShardedInvLists:
    """One shard of a CSR set of lists per rank + gather of decoded ids.

    encode_fn(local_offsets uint64 numpy, local_ids) -> codec object with
    decode_lists(local_list_nos) -> (ids tensor, out_offsets)   [RocLists / EfLists / PackedLists have exactly this]
    `ids` may be a host array (uint64 / int64) or a CUDA int64 tensor (the shard is then cut on the device).
    """

    def __init__(self, offsets, ids, rank, world, encode_fn, group=None, device="cuda"):
        offsets = np.asarray(offsets, dtype=np.uint64)
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.sizes = (offsets[1:] - offsets[:-1]).astype(np.int64)
        self.nlist = self.sizes.size
        self.owner = lpt_partition(self.sizes, world)
        # local numbering: lists of a rank in increasing global number (rank of the list among its owner's lists)
        order = np.argsort(self.owner, kind="stable")
        counts = np.bincount(self.owner, minlength=world).astype(np.int64)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        self.local_no = np.empty(self.nlist, dtype=np.int64)
        self.local_no[order] = np.arange(self.nlist, dtype=np.int64) - np.repeat(first, counts)
        self.my_lists = order[first[rank]:first[rank] + counts[rank]]
        loc_sizes = self.sizes[self.my_lists]
        self.local_offsets = np.concatenate([[0], np.cumsum(loc_sizes)]).astype(np.uint64)
        starts = offsets[:-1].astype(np.int64)[self.my_lists]
        if isinstance(ids, np.ndarray):
            if ids.dtype.kind not in "iu":
                raise TypeError(f"ids must be an integer array, got {ids.dtype}")
            # 8-byte ids are reinterpreted, narrower ones (int32 / uint32 id arrays) converted: a view of those would
            # halve the length and encode garbage
            cut = np.ascontiguousarray(ids[_segment_index(starts, loc_sizes, np)]) if loc_sizes.size else np.zeros(0, dtype=np.uint64)
            local_ids = cut.view(np.uint64) if cut.dtype.itemsize == 8 else cut.astype(np.uint64)
        else:  # CUDA tensor
            import torch

            st = torch.from_numpy(starts).to(ids.device)
            sz = torch.from_numpy(loc_sizes).to(ids.device)
            local_ids = ids[_segment_index(st, sz, torch)] if loc_sizes.size else ids[:0]
        self.codec = encode_fn(self.local_offsets, local_ids)
        self.load = np.bincount(self.owner, weights=self.sizes, minlength=world).astype(np.int64)
        self._p2p_ready = False

    def _ensure_p2p(self):
        """The first operation of a process group must not be a batch of point-to-point operations: PyTorch documents batched
        send / recv as undefined when they are a group's first call (the NCCL / RCCL communicators of the pairs are created
        lazily and at different moments on the two sides).  One collective brings every rank's communicator up first."""
        if self._p2p_ready or self.world <= 1:
            return
        import torch
        import torch.distributed as dist

        t = torch.zeros(1, dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        self._p2p_ready = True

    def decode_local(self, list_nos):
        """Decode the requested GLOBAL list numbers this rank owns (request order, repeats kept)
        -> (their global numbers, ids tensor, offsets of the lists inside it)."""
        ln = np.asarray(list_nos, dtype=np.int64)
        ln = ln[self.owner[ln] == self.rank]
        ids, off = self.codec.decode_lists(self.local_no[ln].astype(np.uint64))
        return ln, ids, off

    def gather_ids(self, list_nos, dst=0):
        """Decoded ids of `list_nos` (global numbers, any owners) assembled on rank `dst`.
        -> (int64 tensor in request order, offsets) on dst, (None, None) elsewhere."""
        import torch
        import torch.distributed as dist

        ln = np.asarray(list_nos, dtype=np.int64)
        sizes = self.sizes[ln]
        req_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        owners = self.owner[ln]
        _, ids, _ = self.decode_local(ln)
        ids = ids.to(self.device).view(torch.int64) if ids.numel() else torch.zeros(0, dtype=torch.int64, device=self.device)
        self._ensure_p2p()
        if self.world > 1 and self.rank != dst:
            if ids.numel():  # (the same batched form as the receiving side: one group call per rank and gather)
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, ids.contiguous(), dst, self.group)]):
                    w.wait()
            return None, None
        out = torch.empty(int(req_off[-1]), dtype=torch.int64, device=self.device)
        bufs = {self.rank: ids}
        if self.world > 1:  # one message per owning rank, sizes known from the replicated map; all receives posted at once
            ops = []
            for r in range(self.world):
                n = int(sizes[owners == r].sum())
                if r == dst or n == 0:
                    continue
                bufs[r] = torch.empty(n, dtype=torch.int64, device=self.device)
                ops.append(dist.P2POp(dist.irecv, bufs[r], r, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        for r, buf in bufs.items():  # the lists of rank r, in request order, to their slots: one indexed copy
            items = np.nonzero(owners == r)[0]
            if items.size == 0:
                continue
            st = torch.from_numpy(req_off[items]).to(self.device)
            sz = torch.from_numpy(sizes[items]).to(self.device)
            out[_segment_index(st, sz, torch)] = buf
        return out, req_off
