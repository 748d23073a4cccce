#!/usr/bin/env python3
"""bench.py -- IDs encoded+decoded per second on the BASELINE.json workload (MI355X).

A "step" = one pass of the hot path over one batch: ROC-encode every inverted list of the batch the way the container
does (stream + the sampling permutation the vector codes are reordered by, custom_invlists_impl.cpp:188-193), then
ROC-decode every list (inputs and outputs resident in HBM).  Default workload = BASELINE.json configs[1]
(S1: 1M uint64 ids in 1024 Zipf(0.75) lists).  With N GPUs every rank owns a shard of the same shape
(different seed): inverted lists are independent, so there is no data-path collective ("weak" scaling).

`--sharded` (with `--workload c5`) is the strong-scaling form of BASELINE.json configs[4]: ONE 65 536-list index,
lists partitioned over the ranks by total length (sharding.ShardedInvLists), every rank encodes + decodes its shard,
then a search-shaped request (nq * nprobe touched lists) is decoded by the owners and gathered on rank 0 over
RCCL send/recv; the line reports the per-rank time spread and the gather time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ROCm runtime setting, read when HIP initialises (i.e. before torch is imported): the number of hardware queues the
# process's streams are multiplexed onto (default 4).  A large ROC call runs up to eight kernel classes side by side
# (vidc_ctx: `wide` when this is >= 8, csrc/common.h); with 4 queues the library falls back to three streams.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(offsets, ids, budget_s=20.0):
    """Time the CPU codec on this host (rank 0, N=1 only).  Test-infrastructure libraries from oracle/."""
    from oracle.pyoracle import Oracle, Ref

    try:
        impl, kind = Ref(), "reference"
    except Exception:
        impl, kind = Oracle(), "port"
    cores = impl.max_threads()
    ntotal = int(offsets[-1])
    t0 = time.time()
    runs = []
    while True:
        r = impl.bench_lists(offsets, ids, cores)
        runs.append(r["t_enc"] + r["t_dec"])
        if time.time() - t0 > budget_s / 2 or len(runs) >= 41:
            break
    t_all = float(np.median(runs))
    ones = [impl.bench_lists(offsets, ids, 1) for _ in range(5)]
    t_one = float(np.median([o["t_enc"] + o["t_dec"] for o in ones]))
    return dict(value=ntotal / t_all, unit="IDs/s (encode+decode)", cores=cores, kind=kind,
                sample=f"full S1 batch ({ntotal} ids / {offsets.size - 1} lists), median of {len(runs)} runs, "
                       f"OpenMP schedule(dynamic) over lists",
                single_thread_value=ntotal / t_one, enc_s=r["t_enc"], dec_s=r["t_dec"],
                bits_per_id=8.0 * r["bytes"] / ntotal, bad_lists=r["bad_lists"])


def committed_traffic(tag, alg_bytes):
    """HBM-side bytes of one step of a workload from the PMC counters: collected offline (rocprofv3 cannot wrap its own caller;
    tools/pmc_workload.sh / tools/pmc_s1.sh: separate FETCH_SIZE / WRITE_SIZE passes of this very command) and committed as
    profiles/pmc_traffic_<tag>.json; tools/final_run.sh regenerates them.  None when the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", f"pmc_traffic_{tag}.json")) as f:
            pt = json.load(f)
        traffic = 1024.0 * (pt["fetch_KiB_per_step"] + pt["write_KiB_per_step"])
        return {"traffic": traffic, "traffic_over_algorithmic": traffic / alg_bytes if alg_bytes else None,
                "fetch_GiB": pt["fetch_KiB_per_step"] / 2 ** 20, "write_GiB": pt["write_KiB_per_step"] / 2 ** 20,
                "source": f"profiles/pmc_traffic_{tag}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, summed over the "
                          "kernels of a step; offline: the counters cannot be read from inside the timed process)"}
    except Exception:
        return None


def per_list_multisets_equal(offsets, got, want, chunk=1 << 27):
    """Every list of `got` holds the same multiset of ids as the same list of `want` (both device int64, CSR `offsets`):
    keyed sort (list number, id) of both sides, in chunks cut on list boundaries.  A list-boundary bug fails this; a
    global sort of the whole array would not see it."""
    import torch

    ends = np.asarray(offsets[1:], dtype=np.int64)
    n = int(ends[-1]) if ends.size else 0
    bounds = torch.from_numpy(ends).to(got.device)
    start = 0
    while start < n:
        j = int(np.searchsorted(ends, min(n, start + chunk), side="right"))
        end = int(ends[j - 1]) if j > 0 and ends[j - 1] > start else int(ends[min(j, ends.size - 1)])
        seg = torch.searchsorted(bounds, torch.arange(start, end, device=got.device), right=True) << 40  # ids < 2^40
        a = torch.sort(seg + got[start:end]).values
        b = torch.sort(seg + want[start:end]).values
        if not torch.equal(a, b):
            return False
        del seg, a, b
        start = end
    return True


def sharded_main(args, ctx, dist, rank, world):
    """Strong scaling of one index (BASELINE configs[4] shape): shard, encode + decode per rank, search-shaped gather."""
    import torch

    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.codecs import RocLists
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    wl = synth.workload(args.workload if args.workload != "s1" else "c5", seed=42)  # the SAME index on every rank
    offsets, ids_host = wl["offsets"], wl["ids"]
    want_perm = not args.no_perm
    t_sh = time.perf_counter()
    sh = ShardedInvLists(offsets, ids_host, rank, world, lambda o, i: (o, torch.from_numpy(np.ascontiguousarray(i).view(np.int64)).cuda()),
                         device="cuda")
    loc_off, loc_ids = sh.codec  # (the "codec" slot holds the raw shard until the first timed encode)
    t_sh = time.perf_counter() - t_sh
    out = torch.empty(int(loc_off[-1]), dtype=torch.int64, device="cuda")
    rng = np.random.default_rng(7)
    nq, nprobe = 1000, 16
    req = rng.integers(0, wl["nlist"], size=nq * nprobe).astype(np.int64)  # lists a batch of searches touched

    def step():
        sh.codec = RocLists.encode(loc_off, loc_ids, ctx=ctx, want_perm=want_perm)
        ke = ctx.phase_ms(0) + ctx.phase_ms(1)
        sh.codec.decode_all(out)
        return ke, ctx.phase_ms(2)

    for _ in range(args.warmup):
        step()
        sh.gather_ids(req, dst=0)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_enc = k_dec = t_codec = t_gather = 0.0
    for _ in range(args.steps):
        ta = time.perf_counter()
        ke, kd = step()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        got, goff = sh.gather_ids(req, dst=0)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        k_enc += ke
        k_dec += kd
        t_codec += tb - ta
        t_gather += tc - tb
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = torch.tensor([t_codec / args.steps, t_gather / args.steps, float(loc_off[-1]), k_enc / args.steps, k_dec / args.steps],
                            dtype=torch.float64, device="cuda")
    allr = [per_rank.clone() for _ in range(world)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_gather(allr, per_rank)
    if rank == 0:
        # the gathered ids are the decoded lists in request order: check them against the index itself (as sets per list)
        sizes = (offsets[1:] - offsets[:-1]).astype(np.int64)
        ok = int(goff[-1]) == int(sizes[req].sum())
        for i in rng.integers(0, req.size, size=64):
            l = int(req[i])
            a = np.sort(got[int(goff[i]):int(goff[i + 1])].cpu().numpy().astype(np.uint64))
            ok = ok and np.array_equal(a, np.sort(np.asarray(ids_host[int(offsets[l]):int(offsets[l + 1])]).astype(np.uint64)))
        rows = [x.cpu().numpy() for x in allr]
        codec_ms = [1e3 * float(x[0]) for x in rows]
        res = {
            "metric": "IDs encoded+decoded / sec (ROC/ANS, bit-exact vs codec.cpp), one index sharded over the GPUs",
            "value": wl["ntotal"] * args.steps / elapsed, "unit": "IDs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl["describe"], "codec": "roc", "nlist": wl["nlist"], "max_list": wl["max_list"],
                       "median_list": wl["median_list"], "container_path": "stream + sampling permutation" if want_perm else "stream only",
                       "parallelism": f"{wl['nlist']} lists partitioned over {world} GPU(s) by total length (LPT); per step: "
                                      f"encode + decode of the shard, then the {nq * nprobe} lists of {nq} searches x nprobe {nprobe} "
                                      f"decoded by their owners and gathered on rank 0 (send/recv, {int(goff[-1]) * 8} bytes)"},
            "per_rank": {"ids": [int(x[2]) for x in rows], "codec_ms": codec_ms, "gather_ms": [1e3 * float(x[1]) for x in rows],
                         "kernel_ms_encode": [float(x[3]) for x in rows], "kernel_ms_decode": [float(x[4]) for x in rows],
                         "codec_ms_spread": max(codec_ms) - min(codec_ms)},
            "shard_setup_s": t_sh, "gather_verified": bool(ok),
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(n):
    """Re-run this command as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same args>`
    (one rank per GPU, rank r bound to GPU r through LOCAL_RANK, nccl = RCCL).  Fails loudly when the node has fewer GPUs."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node")
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = subprocess.call(cmd, env=env)
    if rc:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="s1")
    ap.add_argument("--codec", default="roc", choices=["roc", "ef", "packed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short secondary measurements in `extra`")
    ap.add_argument("--no-s2", action="store_true", help="skip the 1 B-id workload in `extra` (takes ~1 min)")
    ap.add_argument("--sharded", action="store_true", help="strong scaling: one index sharded over the ranks + gather")
    ap.add_argument("--no-perm", action="store_true", help="encode the streams only (no sampling permutation)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL), exactly the
        # command the driver would have used; rank 0 of the children prints the JSON line
        return launch_ranks(args.gpus)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also exercised at world size 1)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from vector_db_id_compression_amd import _lib, synth
    from vector_db_id_compression_amd.codecs import EfLists, PackedLists, RocLists

    ctx = _lib.default_context(local_rank)  # bound to torch's current stream
    if args.sharded:
        return sharded_main(args, ctx, dist, rank, world)
    want_perm = not args.no_perm
    wl = synth.workload(args.workload, seed=42 + rank)
    offsets = wl["offsets"]
    ids_host = wl["ids"] if isinstance(wl["ids"], np.ndarray) else None
    d_ids = torch.from_numpy(ids_host.view(np.int64)).cuda() if ids_host is not None else wl["ids"]
    ntotal = wl["ntotal"]
    out = torch.empty(ntotal, dtype=torch.int64, device="cuda")

    chain_ms = [0.0, 0.0]  # hipEvent time of the launch holding the longest chains (encode, decode), summed over steps

    def step():
        if args.codec == "roc":
            r = RocLists.encode(offsets, d_ids, ctx=ctx, want_perm=want_perm)
            t_enc = ctx.phase_ms(0) + ctx.phase_ms(1)
            chain_ms[0] += ctx.phase_ms(3)
            r.decode_all(out)
            t_dec = ctx.phase_ms(2)
            chain_ms[1] += ctx.phase_ms(4)
        elif args.codec == "ef":
            r = EfLists.encode(offsets, d_ids, ctx=ctx)
            t_enc = ctx.last_kernel_ms()
            r.decode_all(out)
            t_dec = ctx.last_kernel_ms()
        else:
            r = PackedLists.encode(offsets, d_ids, ctx=ctx)
            t_enc = ctx.last_kernel_ms()
            r.decode_all(out)
            t_dec = ctx.last_kernel_ms()
        return r, t_enc, t_dec

    for _ in range(args.warmup):
        r, _, _ = step()
    # correctness gate inside the bench: every list must come back as the same set of ids
    verified = None
    if not args.no_verify:
        verified = per_list_multisets_equal(offsets, out, d_ids)

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    chain_ms[0] = chain_ms[1] = 0.0
    t0 = time.perf_counter()
    k_enc = k_dec = 0.0
    for _ in range(args.steps):
        r, te, td = step()
        k_enc += te
        k_dec += td
    torch.cuda.synchronize()
    t_own = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_own = t_own - t0  # this rank's K steps, before the closing barrier
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if verified is not None:  # the output of the LAST timed step too (outside the timed region)
        verified = verified and (per_list_multisets_equal(offsets, out, d_ids) if args.codec == "roc" else bool(torch.equal(out, d_ids)))

    dominant = None
    if args.codec == "roc" and chain_ms[0] > 0:
        # The dominant kernel by the contract's definition: ONE launch (k_roc_encode_u2<UB, perm>) holding the lists longer than
        # 4096 ids.  Algorithmic bytes of that launch = ids it read (8 B each) + streams and permutation it wrote.
        try:
            ci, cd = ctx.chain_info(0), ctx.chain_info(1)
            info = r.info()
            sizes = (offsets[1:] - offsets[:-1]).astype(np.int64)
            in_launch = (sizes > 4096) & (info["precision"] <= ci["universe_bits"]) & (info["precision"] > (18 if ci["universe_bits"] == 20 else 0))
            stream_bytes = float((8 + 4 * info["nwords"][in_launch].astype(np.int64)).sum())
            n_ids = float(sizes[in_launch].sum())
            enc_bytes = 8.0 * n_ids + stream_bytes + (4.0 * n_ids if want_perm else 0.0)
            dec_bytes = stream_bytes + 8.0 * n_ids
            e_ms, d_ms = chain_ms[0] / args.steps, chain_ms[1] / args.steps
            dominant = {
                "name": f"k_roc_encode_u2<{ci['universe_bits']}, {'true' if want_perm else 'false'}>", "launch_ms": e_ms,
                "lists": ci["lists"], "ids": ci["ids"], "longest_list": ci["longest"], "ids_recount": n_ids,
                "algorithmic_bytes_per_launch": enc_bytes, "achieved_GBs": enc_bytes / e_ms / 1e6, "frac": enc_bytes / e_ms / 1e6 / HBM_PEAK_GBS,
                "second": {"name": f"k_roc_decode_u2<{cd['universe_bits']}>", "launch_ms": d_ms, "ids": cd["ids"],
                           "algorithmic_bytes_per_launch": dec_bytes, "achieved_GBs": dec_bytes / d_ms / 1e6 if d_ms else None,
                           "frac": dec_bytes / d_ms / 1e6 / HBM_PEAK_GBS if d_ms else None},
                "note": "one wavefront per list: the launch lasts as long as its longest list's chain of dependent codec steps"}
        except Exception as e:
            dominant = {"error": str(e)}

    def chain_floor(w2, ids2):
        """The longest list of the workload encoded + decoded ALONE: its serial chain (one dependent codec step per id)
        bounds any schedule of the batch from below."""
        sizes = (w2["offsets"][1:] - w2["offsets"][:-1]).astype(np.int64)
        l = int(np.argmax(sizes))
        a, b = int(w2["offsets"][l]), int(w2["offsets"][l + 1])
        one = ids2[a:b].contiguous()
        off1 = np.array([0, b - a], dtype=np.uint64)
        o1 = torch.empty(b - a, dtype=torch.int64, device="cuda")
        e = d = 0.0
        for it in range(4):
            obj = RocLists.encode(off1, one, ctx=ctx, want_perm=want_perm)
            e_ms = ctx.phase_ms(0)
            obj.decode_all(o1)
            if it:
                e += e_ms
                d += ctx.phase_ms(2)
        return {"list": l, "ids": b - a, "encode_ms": e / 3, "decode_ms": d / 3,
                "us_per_step": {"encode": 1e3 * e / 3 / (b - a), "decode": 1e3 * d / 3 / (b - a)}}

    def secondary(workload, codec, steps=5, floor=False, traffic_tag=None):
        """Short, untimed-by-the-driver measurement of another regime / codec (reported under `extra` only).
        `workload`: a name or an already generated workload dict (the 1 B-id set is generated once for the three codecs)."""
        w2 = synth.workload(workload, seed=1042 + rank) if isinstance(workload, str) else workload
        ids2 = torch.from_numpy(w2["ids"].view(np.int64)).cuda() if isinstance(w2["ids"], np.ndarray) else w2["ids"]
        out2 = torch.empty(w2["ntotal"], dtype=torch.int64, device="cuda")
        cls = {"roc": RocLists, "ef": EfLists, "packed": PackedLists}[codec]
        kw = {"want_perm": want_perm} if codec == "roc" else {}
        ke_l, kd_l, wall_l = [], [], []
        ok, first = True, None
        for it in range(steps + 2):
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            obj = cls.encode(w2["offsets"], ids2, ctx=ctx, **kw)
            e_ms = (ctx.phase_ms(0) + ctx.phase_ms(1)) if codec == "roc" else ctx.last_kernel_ms()
            obj.decode_all(out2)
            d_ms = ctx.phase_ms(2) if codec == "roc" else ctx.last_kernel_ms()
            torch.cuda.synchronize()
            if it >= 2:  # two warm-up passes: the second one still grows the block cache (the previous object is alive while
                # the next one is encoded, so two sets of buffers exist from then on; 267 vs 85 ms per encode call on S2)
                wall_l.append(time.perf_counter() - t_a)
                ke_l.append(e_ms)
                kd_l.append(d_ms)
        # every step is timed on its own and the MEDIAN step is reported (the mean next to it): one hiccup of the box in three
        # steps of a millisecond each otherwise decides the line
        t_wall = float(np.median(wall_l)) * steps
        ke = float(np.median(ke_l)) * steps
        kd = float(np.median(kd_l)) * steps
        # correctness: the last timed pass and three more passes of the same call sequence (outside the timed loop, so that the
        # check's own sorts and copies do not evict the inputs between timed passes).  The first checked ROC decode list by list
        # -- ROC returns every list as the same SET of ids in sampling order --, the others against it (same input -> same
        # streams -> same decoded order); Elias-Fano / packed bits return the input itself.
        for v in range(4):
            if v:
                obj = cls.encode(w2["offsets"], ids2, ctx=ctx, **kw)
                obj.decode_all(out2)
            if codec != "roc":
                ok = ok and bool(torch.equal(out2, ids2))
            elif first is None:
                ok = ok and per_list_multisets_equal(w2["offsets"], out2, ids2)
                first = out2.clone()
            else:
                ok = ok and bool(torch.equal(out2, first))
        del first
        c2 = obj.compressed_bytes / w2["ntotal"]
        kern = (ke + kd) / steps / 1e3
        gbs = (16.0 + 2.0 * c2) * w2["ntotal"] / kern / 1e9
        res2 = {"workload": w2["describe"], "codec": codec, "nlist": w2["nlist"], "max_list": w2["max_list"],
                "median_list": w2["median_list"], "ids_per_s": w2["ntotal"] * steps / t_wall,
                "ms_per_step": 1e3 * t_wall / steps, "kernel_ms": {"encode": ke / steps, "decode": kd / steps},
                "timing": f"median of {steps} separately timed steps", "ms_per_step_mean": 1e3 * float(np.mean(wall_l)),
                "kernel_ms_mean": {"encode": float(np.mean(ke_l)), "decode": float(np.mean(kd_l))},
                # wall clock of a step minus the hipEvent time of its kernels: host planning, launches, synchronisations
                "host_ms_per_step": 1e3 * t_wall / steps - (ke + kd) / steps,
                "bits_per_id": 8.0 * c2, "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                "per_list_roundtrip_ok": ok, "passes_checked": 4}
        if traffic_tag:
            alg2 = (16.0 + 2.0 * c2 + (4.0 if (codec == "roc" and want_perm) else 0.0)) * w2["ntotal"]
            tr2 = committed_traffic(traffic_tag, alg2)
            res2["algorithmic_bytes_per_step"] = alg2
            res2["traffic"] = tr2["traffic"] if tr2 else None
            res2["traffic_over_algorithmic"] = tr2["traffic_over_algorithmic"] if tr2 else None
            if tr2:
                res2["traffic_source"] = tr2["source"]
        if floor:
            del out2
            cf = chain_floor(w2, ids2)
            cf["chain_floor_ms"] = cf["encode_ms"] + cf["decode_ms"]
            cf["share_of_kernel_time"] = cf["chain_floor_ms"] / ((ke + kd) / steps)
            cf["note"] = ("the longest list alone: every codec step consumes the ANS head of the previous one, so no "
                          "schedule of this batch runs faster than this chain (under load its steps are slower: its one "
                          "global load per step misses L2)")
            res2["chain_floor"] = cf
        return res2

    def secondary_graph(N=262144, K=64, steps=3):
        """BASELINE configs[3] shape (NSG adjacency rows, -1 terminated) through the ROC and Elias-Fano graph codecs."""
        rows = torch.from_numpy(synth.make_graph_rows(N, K, seed=1044 + rank)).cuda()
        nodes = np.arange(N, dtype=np.uint64)
        edges = int((rows >= 0).sum().item())
        res = {"workload": f"{N} graph nodes x K={K} int32 rows ({edges} edges)"}
        for name, cls in (("roc", RocLists), ("elias_fano", EfLists)):
            t_wall = ke = kd = 0.0
            for it in range(steps + 1):
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                g = cls.encode_rows(rows, ctx=ctx)
                e_ms = ctx.last_kernel_ms()
                dec, _ = g.decode_rows(None, K, want_counts=False)
                d_ms = ctx.last_kernel_ms()
                torch.cuda.synchronize()
                if it:
                    t_wall += time.perf_counter() - t_a
                    ke += e_ms
                    kd += d_ms
            ok = bool(((dec >= 0).sum() == edges).item())
            res[name] = {"edges_per_s": edges * steps / t_wall, "ms_per_step": 1e3 * t_wall / steps,
                         "kernel_ms": {"encode": ke / steps, "decode": kd / steps},
                         "host_ms_per_step": 1e3 * t_wall / steps - (ke + kd) / steps,
                         "bits_per_edge": 8.0 * g.compressed_bytes / edges, "edge_count_ok": ok}
        return res

    comp_bytes = r.compressed_bytes
    c = comp_bytes / ntotal  # compressed bytes per id
    # enc 8 B read + c written, dec c read + 8 B written; the container path also writes the 4-byte
    # position of every sampled id (custom_invlists_impl.cpp:188-193)
    alg_per_id = 16.0 + 2.0 * c + (4.0 if (want_perm and args.codec == "roc") else 0.0)
    alg_bytes = alg_per_id * ntotal
    kern_s = (k_enc + k_dec) / args.steps / 1e3
    achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0

    # HBM-side traffic of one step from the PMC counters: collected offline (rocprofv3 cannot wrap its own caller),
    # separate FETCH_SIZE / WRITE_SIZE passes of this very command (tools/pmc_s1.sh -> profiles/pmc_traffic_s1.json)
    traffic = None
    traffic_note = None
    if args.codec == "roc" and args.workload == "s1":
        pt = committed_traffic("s1", alg_bytes)
        if pt:
            traffic = pt["traffic"]
            traffic_note = ("bytes per step, FETCH_SIZE + WRITE_SIZE summed over the ROC kernels of a step (rocprofv3 --pmc, "
                            "separate passes, raw counters: gfx950 counts wide coalesced reads at half their bytes)")

    # per-rank step time (weak scaling): what the ">= 6x at 8 GPUs" target is read from the day a multi-GPU run exists
    per_rank_ms = [1e3 * elapsed_own / args.steps]
    if dist is not None and world > 1:
        mine = torch.tensor([per_rank_ms[0]], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x.item()) for x in allr]

    if rank == 0:
        res = {
            "metric": "IDs encoded+decoded / sec (ROC/ANS, bit-exact vs codec.cpp)" if args.codec == "roc"
            else f"IDs encoded+decoded / sec ({args.codec})",
            "value": world * ntotal * args.steps / elapsed,
            "unit": "IDs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": wl["describe"], "codec": args.codec, "ids_per_gpu": ntotal, "lists_per_gpu": wl["nlist"],
                       "max_list": wl["max_list"], "median_list": wl["median_list"],
                       "parallelism": f"lists sharded over {world} GPU(s), no data-path collective",
                       "container_path": "stream + sampling permutation" if want_perm else "stream only"},
            "bits_per_id": 8.0 * c,
            "verified_roundtrip": verified,
            "kernel_ms": {"encode": k_enc / args.steps, "decode": k_dec / args.steps},
            "host_ms_per_step": 1e3 * elapsed / args.steps - (k_enc + k_dec) / args.steps,
            # every rank's own K steps (max over ranks + the closing barrier = ms_per_step): load balance of the weak form
            "per_rank": {"ms_per_step": per_rank_ms, "spread_ms": max(per_rank_ms) - min(per_rank_ms),
                         "class_streams": ctx.class_streams()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "formula": "achieved = algorithmic_bytes_per_id x ids / (encode + compaction + decode kernel time, hipEvents on the "
                                    "library's streams); algorithmic_bytes_per_id = 16 + 2 c (SURVEY 8d: ids read + written, stream "
                                    "written + read, c = compressed bytes per id)" + (" + 4 (the sampling permutation the container path writes)"
                                                                                       if (want_perm and args.codec == "roc") else ""),
                         "kernels": "k_roc_encode_* + k_roc_compact + k_roc_decode_*" if args.codec == "roc" else args.codec,
                         "algorithmic_bytes_per_id": alg_per_id},
        }
        if dominant is not None:
            res["roofline"]["dominant_kernel"] = dominant
        if world == 1 and not args.no_extra and args.codec == "roc" and args.workload == "s1":
            # other regimes of the same kernels, for context only (never part of `value`): many equal lists
            # (no long serial chain) and the two bandwidth-bound codecs of the same plugin surface
            try:
                res["extra"] = {
                    "roc_many_equal_lists": secondary("uniform_16m", "roc", traffic_tag="uniform_16m_roc"),
                    "packed_bits": secondary("uniform_16m", "packed"),
                    "elias_fano": secondary("uniform_16m", "ef"),
                    "graph_rows": secondary_graph(),
                    "c5": secondary("c5", "roc", floor=True),
                    # the size a search issues (S1: 1 M ids / 1024 lists): launch-bound, a few kernels of ~10 us each
                    "s1_elias_fano": secondary(wl, "ef"),
                    "s1_packed_bits": secondary(wl, "packed"),
                }
                if not args.no_s2:  # BASELINE north_star's roofline workload: 1 B ids on one GPU, through the three codecs
                    torch.cuda.empty_cache()
                    ws2 = synth.workload("s2", seed=1042 + rank)
                    res["extra"]["s2"] = secondary(ws2, "roc", steps=3, floor=True, traffic_tag="s2_roc")
                    res["extra"]["s2_elias_fano"] = secondary(ws2, "ef", steps=3, traffic_tag="s2_ef")
                    res["extra"]["s2_packed_bits"] = secondary(ws2, "packed", steps=3, traffic_tag="s2_packed")
                    del ws2
            except Exception as e:
                res["extra"] = {"error": str(e)}
        if world == 1 and args.codec == "roc" and not args.no_extra:
            try:
                cf = chain_floor(wl, d_ids)
                cf["chain_floor_ms"] = cf["encode_ms"] + cf["decode_ms"]
                cf["share_of_kernel_time"] = cf["chain_floor_ms"] / (1e3 * kern_s)
                res["roofline"]["chain_floor"] = cf
                res["roofline"]["note"] = ("latency bound, not bandwidth bound: the batch cannot finish before its longest list's "
                                           "chain of dependent codec steps (chain_floor); frac is reported against the HBM peak as "
                                           "the contract asks")
            except Exception as e:
                res["roofline"]["chain_floor"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and args.codec == "roc" and ids_host is not None:
            try:
                res["cpu_baseline"] = cpu_baseline(offsets, ids_host)
            except Exception as e:  # the checker libraries are optional at bench time
                res["cpu_baseline"] = {"value": None, "unit": "IDs/s (encode+decode)", "cores": 0, "kind": "port",
                                       "sample": f"unavailable: {e}"}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
