"""A small IVF index used as the *caller* of the ID codecs where Faiss is not installed.

It reproduces the call pattern the reference exercises through Faiss:
  * `search`                      -> scan the probed lists, ids via `invlists.get_ids` of every probed list
                                     (IndexIVF::search_preassigned with store_pairs = false)
  * `search_defer_id_decoding`    -> scan with (list_no << 32 | offset) labels, translate afterwards, either one
                                     by one (`get_single_id`, custom_invlists_impl.cpp:466-475) or per touched list
                                     (`get_ids`, :477-525) -- here as ONE batched device decode
The vector side (k-means, flat / PQ codes, distance scan) is plain torch: it is harness plumbing, not the
product.  Codes: "Flat" (float32 bytes) or ("PQ", M) with 8-bit sub-quantizers (PQ4np / PQ16 style).
"""
import numpy as np

from .invlists import ArrayInvertedLists, to_csr


def _torch():
    import torch

    return torch


def _assign(x, c, chunk=1 << 16):
    """argmin_j ||x_i - c_j|| in chunks (keeps the distance matrix small)."""
    torch = _torch()
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    for a in range(0, x.shape[0], chunk):
        out[a:a + chunk] = torch.cdist(x[a:a + chunk], c).argmin(1)
    return out


def _kmeans(x, k, iters=10, seed=123):
    torch = _torch()
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    n = x.shape[0]
    perm = torch.randperm(n, generator=g)[:k].to(x.device)
    c = x[perm].clone()
    if c.shape[0] < k:  # fewer points than centroids: pad with jittered copies
        reps = (k + c.shape[0] - 1) // c.shape[0]
        c = c.repeat(reps, 1)[:k] + 1e-3 * torch.arange(k, device=x.device, dtype=x.dtype)[:, None]
    for _ in range(iters):
        a = _assign(x, c)
        sums = torch.zeros_like(c).index_add_(0, a, x)
        cnt = torch.bincount(a, minlength=k).to(x.dtype)[:, None]
        c = torch.where(cnt > 0, sums / cnt.clamp(min=1), c)
    return c


class IVFIndex:
    def __init__(self, d, nlist, code="Flat"):
        self.d, self.nlist = int(d), int(nlist)
        self.code = code
        self.nprobe = 1
        self.parallel_mode = 0
        self.centroids = None
        self.pq = None  # [M, 256, dsub]
        self.M = code[1] if isinstance(code, tuple) else 0
        self.code_size = 4 * self.d if code == "Flat" else self.M
        self.invlists = ArrayInvertedLists(self.nlist, self.code_size)
        self.ntotal = 0
        self._dev = None

    # -- build
    def train(self, xt):
        torch = _torch()
        xt = torch.as_tensor(np.asarray(xt, dtype=np.float32)).cuda()
        if xt.shape[0] > 200_000:  # like Faiss, train on a subsample
            g = torch.Generator(device="cpu")
            g.manual_seed(99)
            xt = xt[torch.randperm(xt.shape[0], generator=g)[:200_000].to(xt.device)]
        self.centroids = _kmeans(xt, self.nlist)
        if self.M:
            dsub = self.d // self.M
            assert dsub * self.M == self.d
            self.pq = torch.stack([_kmeans(xt[:, m * dsub:(m + 1) * dsub], 256, seed=7 + m) for m in range(self.M)])

    def _encode(self, x):
        torch = _torch()
        if not self.M:
            return x.contiguous().cpu().numpy().view(np.uint8).reshape(x.shape[0], 4 * self.d)
        dsub = self.d // self.M
        codes = torch.stack([_assign(x[:, m * dsub:(m + 1) * dsub].contiguous(), self.pq[m]) for m in range(self.M)], 1)
        return codes.to(torch.uint8).cpu().numpy()

    def add(self, xb):
        torch = _torch()
        xb = torch.as_tensor(np.asarray(xb, dtype=np.float32)).cuda()
        assign = _assign(xb, self.centroids).cpu().numpy()
        codes = self._encode(xb)
        ids = np.arange(self.ntotal, self.ntotal + xb.shape[0], dtype=np.int64)
        order = np.argsort(assign, kind="stable")
        bounds = np.searchsorted(assign[order], np.arange(self.nlist + 1))
        for l in range(self.nlist):
            sel = order[bounds[l]:bounds[l + 1]]
            if sel.size:
                self.invlists.add_entries(l, ids[sel], codes[sel])
        self.ntotal += xb.shape[0]
        self._dev = None

    def replace_invlists(self, il, own=False):
        self.invlists = il
        self._dev = None

    # -- device view of the codes of the current invlists
    def _device_lists(self):
        torch = _torch()
        if self._dev is None:
            il = self.invlists
            if getattr(il, "codes_all", None) is not None:  # compressed container: codes already resident
                off, codes = il._offsets, il.codes_all
            else:
                off, _, codes = to_csr(il)
                codes = torch.from_numpy(codes).cuda()
            self._dev = (np.asarray(off, dtype=np.int64), codes)
        return self._dev

    def _decode_codes(self, codes):
        torch = _torch()
        if not self.M:
            return codes.contiguous().view(torch.float32).reshape(-1, self.d)
        dsub = self.d // self.M
        parts = [self.pq[m][codes[:, m].long()] for m in range(self.M)]
        return torch.cat(parts, 1).reshape(-1, dsub * self.M)

    def _scan(self, xq, k):
        """-> D [nq,k] float32, labels [nq,k] int64 = list_no << 32 | offset (-1 when fewer than k candidates)."""
        torch = _torch()
        off, codes = self._device_lists()
        xq = torch.as_tensor(np.asarray(xq, dtype=np.float32)).cuda()
        nq = xq.shape[0]
        probe = torch.cdist(xq, self.centroids).topk(min(self.nprobe, self.nlist), largest=False).indices.cpu().numpy()
        D = np.full((nq, k), np.inf, dtype=np.float32)
        L = np.full((nq, k), -1, dtype=np.int64)
        for q in range(nq):
            rows, labs = [], []
            for l in probe[q]:
                a, b = int(off[l]), int(off[l + 1])
                if b > a:
                    rows.append(torch.arange(a, b, device="cuda"))
                    labs.append((int(l) << 32) + torch.arange(b - a, device="cuda", dtype=torch.int64))
            if not rows:
                continue
            rows, labs = torch.cat(rows), torch.cat(labs)
            vecs = self._decode_codes(codes[rows])
            dist = ((vecs - xq[q][None, :]) ** 2).sum(1)
            kk = min(k, dist.numel())
            top = torch.topk(dist, kk, largest=False, sorted=True)
            D[q, :kk] = top.values.cpu().numpy()
            L[q, :kk] = labs[top.indices].cpu().numpy()
        return D, L

    # -- searches
    def search(self, xq, k):
        """Non-deferred path: every probed list is fully decoded (get_ids) like IndexIVF::search_preassigned."""
        D, L = self._scan(xq, k)
        I = np.full_like(L, -1)
        valid = L >= 0
        lists = (L[valid] >> 32).astype(np.int64)
        offs = (L[valid] & 0xFFFFFFFF).astype(np.int64)
        cache = {}
        out = np.zeros(lists.size, np.int64)
        for i, (l, o) in enumerate(zip(lists, offs)):
            if l not in cache:
                cache[l] = np.asarray(self.invlists.get_ids(int(l)))
            out[i] = cache[l][o]
        I[valid] = out
        return D, I

    def search_defer_id_decoding(self, xq, k, decode_1by1=False, return_codes=0):
        """custom_invlists_impl.cpp:407-526 (parallel_mode == 3 is required there, :420-422)."""
        torch = _torch()
        if self.parallel_mode != 3:
            raise RuntimeError("set the parallel mode to 3 otherwise search will be single-threaded")
        D, L = self._scan(xq, k)
        I = np.full_like(L, -1)
        valid = L >= 0
        lists = (L[valid] >> 32).astype(np.uint64)
        offs = (L[valid] & 0xFFFFFFFF).astype(np.uint64)
        il = self.invlists
        if decode_1by1:
            if hasattr(il, "get_single_ids"):
                I[valid] = il.get_single_ids(lists, offs)
            else:
                I[valid] = [il.get_single_id(int(l), int(o)) for l, o in zip(lists, offs)]
        else:
            uniq, inv = np.unique(lists, return_inverse=True)  # touched lists (:477-502)
            if hasattr(il, "decode_gather"):  # ONE library call: batched device decode + device-side scatter (:508-525)
                I[valid] = il.decode_gather(uniq, inv, offs)
            elif hasattr(il, "decode_lists"):  # a container with a batched decode only: index on the device with torch
                ids, out_off = il.decode_lists(uniq)
                pos = torch.from_numpy((out_off[inv].astype(np.int64) + offs.astype(np.int64))).cuda()
                I[valid] = ids[pos].cpu().numpy()
            else:
                per = [np.asarray(il.get_ids(int(l))) for l in uniq]
                I[valid] = [per[i][int(o)] for i, o in zip(inv, offs)]
        if not return_codes:
            return D, I
        off, codes = self._device_lists()
        cs1 = self.code_size + (1 if return_codes == 2 else 0)  # coarse_code_size() = 1 byte for nlist <= 256
        out = np.full(L.shape + (cs1,), 0xFF, dtype=np.uint8)  # memset(code1, -1, code_size_1), :444-446
        rows = off[lists.astype(np.int64)] + offs.astype(np.int64)
        got = codes[torch.from_numpy(rows).cuda()].cpu().numpy()
        if return_codes == 2:
            got = np.concatenate([lists.astype(np.uint8)[:, None], got], 1)  # encode_listno, :455-458
        out[valid] = got
        return D, I, out
