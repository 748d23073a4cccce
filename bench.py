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

With `--gpus N > 1` the ONE launch (one process group) answers both multi-GPU questions: the weak aggregate as `value`, and
`extra.sharded_c5` = the strong-scaling form run right behind it by the same ranks (per-rank codec ms, spread, gather ms and
bytes); `rccl` names the backend, world size and devices the collectives ran on.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.  The line is kept short
enough for the driver's log tail (numbers rounded to 5 significant digits, `extra` entries under short keys: `extra_legend`);
the unabridged measurements go to gpurun_out/bench_detail.json.

`--dry-run` (tests/test_bench_cpu.py, no GPU): the same control flow, process group (gloo) and line assembly with a
pass-through stand-in for the codec; it prints `"dry_run": true` and `"value": null` -- it measures nothing.
"""
import argparse
import json
import os
import re
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


def sig(x, n=5):
    """Numbers of the printed line rounded to n significant digits (recursively); everything else untouched."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        return float(f"{x:.{n}g}") if np.isfinite(x) else None
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return x


EXTRA_LEGEND = ("per entry: ids_per_s, ms = wall ms per encode+decode step (median), k_ms = [encode, decode] kernel ms (hipEvents), "
                "host_ms = ms - sum(k_ms), bits = bits/id, frac = (16+2c) B/id x ids / kernel time / 8 TB/s, frac_wall = the same over "
                "ms, traffic_x = PMC FETCH+WRITE bytes / algorithmic bytes (profiles/pmc_traffic_*.json), chain_ms = longest list "
                "alone [encode, decode], ok = per-list round trip of 4 passes")


def compact(r2):
    """A `secondary()` result under short keys (the driver keeps the last ~8000 characters of the line)."""
    if not isinstance(r2, dict) or "kernel_ms" not in r2:
        return r2
    c = {"ids_per_s": r2["ids_per_s"], "ms": r2["ms_per_step"], "k_ms": [r2["kernel_ms"]["encode"], r2["kernel_ms"]["decode"]],
         "host_ms": r2["host_ms_per_step"], "bits": r2["bits_per_id"], "frac": r2["frac_of_hbm_peak"],
         "frac_wall": r2["frac_of_hbm_peak_wall"], "nlist": r2["nlist"], "max_list": r2["max_list"], "ok": r2["per_list_roundtrip_ok"]}
    if r2.get("traffic_over_algorithmic") is not None:
        c["traffic_x"] = r2["traffic_over_algorithmic"]
    if "chain_floor" in r2:
        c["chain_ms"] = [r2["chain_floor"]["encode_ms"], r2["chain_floor"]["decode_ms"]]
    if r2.get("issue"):
        c["issue"] = {"valu_floor_ms": r2["issue"]["valu_floor_ms"], "frac": r2["issue"]["frac"],
                      "class_alone_ms": r2["issue"].get("class_alone_ms")}
    return c


def write_detail(res):
    """The unabridged measurements next to the printed line (scratch directory of the GPU box; never part of the contract)."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
            json.dump(res, f)
        return "gpurun_out/bench_detail.json"
    except Exception:
        return None


class DryLists:
    """`--dry-run` only (tests/test_bench_cpu.py, no GPU): a pass-through stand-in with the codec objects' call shape, so that the
    bench's control flow, process group and line assembly can be exercised on CPU.  It encodes nothing and the line says so."""

    def __init__(self, offsets, ids):
        import torch

        self.offsets = np.asarray(offsets, dtype=np.uint64)
        self.ids = ids if hasattr(ids, "numel") else torch.from_numpy(np.ascontiguousarray(ids).view(np.int64))
        self.compressed_bytes = 8 * int(self.offsets[-1])

    def decode_all(self, out):
        out.copy_(self.ids)
        return out

    def decode_lists(self, list_nos):
        import torch

        ln = np.asarray(list_nos, dtype=np.int64)
        a, b = self.offsets[ln].astype(np.int64), self.offsets[ln + 1].astype(np.int64)
        parts = [self.ids[int(x):int(y)] for x, y in zip(a, b)]
        off = np.concatenate([[0], np.cumsum(b - a)]).astype(np.uint64)
        return (torch.cat(parts) if parts else self.ids[:0]), off


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


SIMDS, CLOCK_HZ = 1024, 2.4e9  # MI355X: 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles


def committed_issue(tag, kernel_ms):
    """The bound ROC is actually near: instruction issue.  Sum of SQ_INSTS_VALU (wave instructions) over the kernels of one step, from
    the committed counter passes of this very workload (tools/pmc_issue.sh -> profiles/pmc_issue_<tag>.json; offline, like
    committed_traffic), x 4 cycles / (1024 SIMDs x 2.4 GHz) = the time the step's vector instructions need with every SIMD issuing
    one every cycle-slot: `valu_floor_ms`; `frac` = that floor over the measured kernel time.  S2 additionally carries what its two
    chain kernel classes take with the machine to themselves (profiles/r06_s2_timeline_serial.txt, VIDC_SERIAL=1)."""
    try:
        with open(os.path.join(ROOT, "profiles", f"pmc_issue_{tag}.json")) as f:
            pk = json.load(f)["per_kernel_per_step"]
        valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in pk.values())
        salu = sum(v.get("SQ_INSTS_SALU", 0.0) for v in pk.values())
        floor_ms = 1e3 * valu * 4.0 / (SIMDS * CLOCK_HZ)
        out = {"bound": "valu issue", "valu_wave_insts_per_step": valu, "salu_wave_insts_per_step": salu, "valu_floor_ms": floor_ms,
               "frac": floor_ms / kernel_ms if kernel_ms else None,
               "source": f"profiles/pmc_issue_{tag}.json (rocprofv3 --pmc SQ_INSTS_VALU ..., offline passes of this command)"}
        top = sorted(pk.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0.0))[:3]
        out["top_kernels_valu_share"] = {k: v.get("SQ_INSTS_VALU", 0.0) / valu for k, v in top} if valu else {}
        if tag.startswith("s2"):
            alone = {}
            with open(os.path.join(ROOT, "profiles", "r06_s2_timeline_serial.txt")) as f:
                for line in f:
                    m = re.match(r"^(k_roc_encode_r2|k_roc_decode_b2)\b.*start ([\d.]+) end ([\d.]+)", line)
                    if m:
                        alone[m.group(1)] = alone.get(m.group(1), 0.0) + float(m.group(3)) - float(m.group(2))
            out["class_alone_ms"] = alone
        return out
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


def sharded_measure(args, ctx, dist, rank, world, steps, warmup, workload="c5", device="cuda", codec="roc"):
    """Strong scaling of one index (BASELINE configs[4] shape): shard, encode + decode per rank, search-shaped gather.
    -> the result dict on rank 0, None elsewhere.  `ctx` None = --dry-run (DryLists stands in for the codec, device cpu)."""
    import torch

    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    wl = synth.workload(workload, seed=42)  # the SAME index on every rank
    offsets, ids_host = wl["offsets"], wl["ids"]
    want_perm = not args.no_perm
    dry = ctx is None
    if not dry:
        from vector_db_id_compression_amd.codecs import EfLists, PackedLists, RocLists

    def sync():
        if not dry:
            torch.cuda.synchronize()

    t_sh = time.perf_counter()
    sh = ShardedInvLists(offsets, ids_host, rank, world,
                         lambda o, i: (o, torch.from_numpy(np.ascontiguousarray(i).view(np.int64)).to(device)), device=device)
    loc_off, loc_ids = sh.codec  # (the "codec" slot holds the raw shard until the first timed encode)
    t_sh = time.perf_counter() - t_sh
    out = torch.empty(int(loc_off[-1]), dtype=torch.int64, device=device)
    rng = np.random.default_rng(7)
    nq, nprobe = 1000, 16
    req = rng.integers(0, wl["nlist"], size=nq * nprobe).astype(np.int64)  # lists a batch of searches touched

    def step():
        if dry:
            sh.codec = DryLists(loc_off, loc_ids)
            sh.codec.decode_all(out)
            return 0.0, 0.0
        if codec == "roc":
            sh.codec = RocLists.encode(loc_off, loc_ids, ctx=ctx, want_perm=want_perm)
            ke = ctx.phase_ms(0) + ctx.phase_ms(1)
            sh.codec.decode_all(out)
            return ke, ctx.phase_ms(2)
        sh.codec = (EfLists if codec == "ef" else PackedLists).encode(loc_off, loc_ids, ctx=ctx)
        ke = ctx.last_kernel_ms()
        sh.codec.decode_all(out)
        return ke, ctx.last_kernel_ms()

    if dist is not None:
        dist.barrier()  # (a collective before the first gather: batched send / recv must not be a group's first operation)
    for _ in range(max(warmup, 1)):
        step()
        sh.gather_ids(req, dst=0)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    k_enc = k_dec = t_codec = t_gather = 0.0
    for _ in range(steps):
        ta = time.perf_counter()
        ke, kd = step()
        sync()
        tb = time.perf_counter()
        got, goff = sh.gather_ids(req, dst=0)
        sync()
        tc = time.perf_counter()
        k_enc += ke
        k_dec += kd
        t_codec += tb - ta
        t_gather += tc - tb
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = torch.tensor([t_codec / steps, t_gather / steps, float(loc_off[-1]), k_enc / steps, k_dec / steps],
                            dtype=torch.float64, device=device)
    allr = [per_rank.clone() for _ in range(world)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_gather(allr, per_rank)
    if rank != 0:
        return None
    # the gathered ids are the decoded lists in request order: check them against the index itself (as sets per list)
    sizes = (offsets[1:] - offsets[:-1]).astype(np.int64)
    ok = int(goff[-1]) == int(sizes[req].sum())
    for i in rng.integers(0, req.size, size=64):
        l = int(req[i])
        a = np.sort(got[int(goff[i]):int(goff[i + 1])].cpu().numpy().astype(np.uint64))
        ok = ok and np.array_equal(a, np.sort(np.asarray(ids_host[int(offsets[l]):int(offsets[l + 1])]).astype(np.uint64)))
    rows = [x.cpu().numpy() for x in allr]
    codec_ms = [1e3 * float(x[0]) for x in rows]
    # what strong scaling can reach at most: the ranks work in parallel, a list does not -- ROC's longest list is one serial chain
    # (sum of ids / (G x longest list)); the bandwidth codecs split a list over wavefronts and have no such bound
    strong_bound = float(wl["ntotal"]) / (world * max(1, wl["max_list"])) if codec == "roc" else None
    return {
        "metric": ("IDs encoded+decoded / sec (ROC/ANS, bit-exact vs codec.cpp), one index sharded over the GPUs" if codec == "roc"
                   else f"IDs encoded+decoded / sec ({codec}), one index sharded over the GPUs"),
        "strong_bound": strong_bound,
        "value": None if dry else wl["ntotal"] * steps / elapsed, "unit": "IDs/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": wl["describe"], "codec": codec, "nlist": wl["nlist"], "max_list": wl["max_list"],
                   "median_list": wl["median_list"],
                   "container_path": ("stream + sampling permutation" if want_perm else "stream only") if codec == "roc" else "streams",
                   "parallelism": f"{wl['nlist']} lists partitioned over {world} GPU(s) by total length (LPT); per step: "
                                  f"encode + decode of the shard, then the {nq * nprobe} lists of {nq} searches x nprobe {nprobe} "
                                  f"decoded by their owners and gathered on rank 0 (send/recv)"},
        "per_rank": {"ids": [int(x[2]) for x in rows], "codec_ms": codec_ms, "gather_ms": [1e3 * float(x[1]) for x in rows],
                     "kernel_ms_encode": [float(x[3]) for x in rows], "kernel_ms_decode": [float(x[4]) for x in rows],
                     "codec_ms_spread": max(codec_ms) - min(codec_ms)},
        "gather_bytes": int(goff[-1]) * 8, "gather_lists": int(req.size),
        "shard_setup_s": t_sh, "gather_verified": bool(ok),
    }


def rccl_info(dist, world, device):
    """What the collectives of this launch ran on: backend (nccl = RCCL on ROCm), world size, one device name per rank."""
    import torch

    name = "cpu" if device == "cpu" else f"{torch.cuda.get_device_name(torch.cuda.current_device())} (cuda:{torch.cuda.current_device()})"
    names = [name]
    if dist is not None and world > 1:
        names = [None] * world
        dist.all_gather_object(names, name)
    ver = None
    if device != "cpu" and dist is not None:
        try:
            ver = ".".join(map(str, torch.cuda.nccl.version()))
        except Exception:
            ver = None
    return {"world": dist.get_world_size() if dist is not None else 1, "backend": dist.get_backend() if dist is not None else None,
            "devices": names, "rccl_version": ver}


def sharded_main(args, ctx, dist, rank, world, device="cuda"):
    """`--sharded`: the strong-scaling form as the printed line."""
    res = sharded_measure(args, ctx, dist, rank, world, args.steps, args.warmup,
                          workload=args.workload if args.workload != "s1" else "c5", device=device)
    info = rccl_info(dist, world, device)
    if rank == 0:
        res["rccl"] = info
        if ctx is None:
            res["dry_run"] = True
        print(json.dumps(sig(res)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(n, dry=False):
    """Re-run this command as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same args>`
    (one rank per GPU, rank r bound to GPU r through LOCAL_RANK, nccl = RCCL).  Fails loudly when the node has fewer GPUs."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if have < n and not dry:
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
    ap.add_argument("--no-sharded-extra", action="store_true", help="with --gpus N > 1: skip extra.sharded_c5")
    ap.add_argument("--sharded-workload", default="c5", help="the ONE index of extra.sharded_c5 (a synth.workload name)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU (tests): gloo, cpu tensors, a pass-through stand-in for the codec; prints value null")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL), exactly the
        # command the driver would have used; rank 0 of the children prints the JSON line
        return launch_ranks(args.gpus, dry=args.dry_run)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = args.dry_run
    device = "cpu" if dry else "cuda"
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also exercised at world size 1)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from vector_db_id_compression_amd import synth

    if dry:
        ctx = None
        RocLists = EfLists = PackedLists = DryLists
    else:
        from vector_db_id_compression_amd import _lib
        from vector_db_id_compression_amd.codecs import EfLists, PackedLists, RocLists

        ctx = _lib.default_context(local_rank)  # bound to torch's current stream
    if args.sharded:
        return sharded_main(args, ctx, dist, rank, world, device=device)
    want_perm = not args.no_perm
    wl = synth.workload(args.workload, seed=42 + rank)
    offsets = wl["offsets"]
    ids_host = wl["ids"] if isinstance(wl["ids"], np.ndarray) else None
    d_ids = torch.from_numpy(ids_host.view(np.int64)).to(device) if ids_host is not None else wl["ids"]
    ntotal = wl["ntotal"]
    out = torch.empty(ntotal, dtype=torch.int64, device=device)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    chain_ms = [0.0, 0.0]  # hipEvent time of the launch holding the longest chains (encode, decode), summed over steps

    def step():
        if dry:  # (stand-in: nothing is encoded)
            r = DryLists(offsets, d_ids)
            r.decode_all(out)
            return r, 0.0, 0.0
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

    for _ in range(max(args.warmup, 1) if dry else args.warmup):
        r, _, _ = step()
    # correctness gate inside the bench: every list must come back as the same set of ids
    verified = None
    if not args.no_verify:
        verified = per_list_multisets_equal(offsets, out, d_ids)

    if dist is not None:
        dist.barrier()
    sync()
    chain_ms[0] = chain_ms[1] = 0.0
    t0 = time.perf_counter()
    k_enc = k_dec = 0.0
    for _ in range(args.steps):
        r, te, td = step()
        k_enc += te
        k_dec += td
    sync()
    t_own = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_own = t_own - t0  # this rank's K steps, before the closing barrier
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
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
                           "frac": dec_bytes / d_ms / 1e6 / HBM_PEAK_GBS if d_ms else None}}
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
                # the same algorithmic bytes over the WALL clock of a step (host planning, launches, waits included)
                "frac_of_hbm_peak_wall": (16.0 + 2.0 * c2) * w2["ntotal"] / (t_wall / steps) / 1e9 / HBM_PEAK_GBS,
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
        if traffic_tag and codec == "roc":
            res2["issue"] = committed_issue(traffic_tag, (ke + kd) / steps)
        return res2

    def secondary_graph(N=1_000_000, K=64, steps=5):
        """BASELINE configs[3] (10^6 nodes x K = 64 int32 rows, -1 terminated) through the three graph containers
        (altid_impl.cpp:20-165).  `frac` = SURVEY 8(d)'s graph variant of the algorithmic bytes, (8 + 2c) B per edge, over the kernels'
        hipEvent time against the 8 TB/s peak; `frac_wall` the same bytes over the wall clock of the two calls; `ok` = every row's
        decoded neighbours, sorted, equal the row's sorted neighbours (compared on the device)."""
        from vector_db_id_compression_amd.codecs import CompactRows
        rows = torch.from_numpy(synth.make_graph_rows(N, K, seed=1044 + rank)).cuda()
        edges = int((rows >= 0).sum().item())
        big = torch.iinfo(torch.int32).max
        want = torch.sort(torch.where(rows >= 0, rows, torch.full_like(rows, big)), dim=1).values
        res = {"workload": f"{N} graph nodes x K={K} int32 rows ({edges} edges)"}
        for name, cls in (("roc", RocLists), ("elias_fano", EfLists), ("compact", CompactRows)):
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
            got = torch.sort(torch.where(dec >= 0, dec, torch.full_like(dec, big)), dim=1).values
            ok = bool(torch.equal(got, want))  # (the data has no power-of-two row maximum above 1: ROC is lossless on it)
            del got
            size = g.size_in_bytes if name == "compact" else g.compressed_bytes
            alg = (8.0 + 2.0 * size / edges) * edges
            res[name] = {"edges_per_s": edges * steps / t_wall, "ms_per_step": 1e3 * t_wall / steps,
                         "kernel_ms": {"encode": ke / steps, "decode": kd / steps},
                         "host_ms_per_step": 1e3 * t_wall / steps - (ke + kd) / steps,
                         "bits_per_edge": 8.0 * size / edges, "algorithmic_bytes_per_step": alg,
                         "frac": alg / ((ke + kd) / steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_wall": alg / (t_wall / steps) / 1e9 / HBM_PEAK_GBS, "roundtrip_ok": ok}
            del g, dec
        return res

    def secondary_wavelet(steps=5):
        """the wavelet-tree container (custom_invlists_impl.cpp:346-397) on 16 M ids in 65 536 lists, plain bit vectors
        (wt_type 0) and RRR-coded ones (wt_type 1): build, decode_all, random selects (get_single_id).  `frac` = (16 + 2c) B/id over
        the build + decode_all kernel time, c = the object's size per id; `ok` = every list decodes to its ids."""
        from vector_db_id_compression_amd.codecs import WaveletTreeLists
        w = synth.workload("uniform_16m", seed=1046 + rank)
        ids = w["ids"]
        if isinstance(ids, np.ndarray):
            ids = torch.from_numpy(ids.view(np.int64)).cuda()
        off, n = w["offsets"], w["ntotal"]
        rng = np.random.default_rng(3)
        ln = rng.integers(0, w["nlist"], 100000).astype(np.uint64)
        sizes = (off[1:] - off[:-1])[ln.astype(np.int64)]
        ln, sizes = ln[sizes > 0], sizes[sizes > 0]
        of = (rng.random(ln.size) * sizes).astype(np.uint64)
        res = {"workload": w["describe"]}
        for wt_type in (0, 1):
            tb = td = kb = kd = ts = 0.0
            for it in range(steps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                wt = WaveletTreeLists.build(off, ids, wt_type=wt_type, ctx=ctx)
                t1 = time.perf_counter()
                k1 = ctx.last_kernel_ms()
                dec = wt.decode_all()
                t2 = time.perf_counter()
                k2 = ctx.last_kernel_ms()
                got = wt.select(ln, of)
                t3 = time.perf_counter()
                if it:
                    tb += t1 - t0; td += t2 - t1; kb += k1; kd += k2; ts += t3 - t2
            ok = bool(torch.equal(dec, ids)) and bool(np.array_equal(got, ids[torch.from_numpy((off[ln.astype(np.int64)] + of).astype(np.int64)).cuda()].cpu().numpy()))
            c = wt.size_in_bytes / n
            alg = (16.0 + 2.0 * c) * n
            res[f"wt_type_{wt_type}"] = {"build_ms": 1e3 * tb / steps, "decode_all_ms": 1e3 * td / steps,
                                          "kernel_ms": {"build": kb / steps, "decode_all": kd / steps},
                                          "selects_per_s": ln.size * steps / ts, "bits_per_id": 8.0 * c,
                                          "frac": alg / ((kb + kd) / steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "frac_wall": alg / ((tb + td) / steps) / 1e9 / HBM_PEAK_GBS, "ok": ok}
            del wt, dec
        return res

    comp_bytes = r.compressed_bytes
    c = comp_bytes / ntotal  # compressed bytes per id
    # enc 8 B read + c written, dec c read + 8 B written = 16 + 2c B/id: that is `roofline.achieved` / `frac`.
    # The container path also writes the 4-byte position of every sampled id (custom_invlists_impl.cpp:188-193), which 8(d)
    # does not define: printed beside it as `frac_with_perm`.
    alg_per_id = 16.0 + 2.0 * c
    perm_per_id = 4.0 if (want_perm and args.codec == "roc") else 0.0
    alg_bytes = alg_per_id * ntotal
    kern_s = (k_enc + k_dec) / args.steps / 1e3
    achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
    achieved_perm = (alg_per_id + perm_per_id) * ntotal / kern_s / 1e9 if kern_s > 0 else 0.0

    # HBM-side traffic of one step from the PMC counters: collected offline (rocprofv3 cannot wrap its own caller),
    # separate FETCH_SIZE / WRITE_SIZE passes of this very command (tools/pmc_s1.sh -> profiles/pmc_traffic_s1.json)
    traffic = None
    if args.codec == "roc" and args.workload == "s1" and not dry:
        pt = committed_traffic("s1", (alg_per_id + perm_per_id) * ntotal)
        if pt:
            traffic = pt["traffic"]

    # per-rank step time (weak scaling): what the ">= 6x at 8 GPUs" target is read from the day a multi-GPU run exists
    per_rank_ms = [1e3 * elapsed_own / args.steps]
    if dist is not None and world > 1:
        mine = torch.tensor([per_rank_ms[0]], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x.item()) for x in allr]

    # one launch answers both multi-GPU questions: behind the weak steps the SAME ranks shard ONE index (strong form)
    sharded = sharded_ef = None
    if world > 1 and not args.no_sharded_extra and args.codec == "roc":
        try:
            sharded = sharded_measure(args, ctx, dist, rank, world, steps=min(args.steps, 5), warmup=1,
                                      workload=args.sharded_workload, device=device)
        except Exception as e:  # (every rank takes the same path: a failure here is reported, not fatal for `value`)
            sharded = {"error": str(e)}
        try:  # the same index through a codec without a serial chain: the form the ">= 6x at 8 GPUs" of north_star can show on
            sharded_ef = sharded_measure(args, ctx, dist, rank, world, steps=min(args.steps, 5), warmup=1,
                                         workload=args.sharded_workload, device=device, codec="ef")
        except Exception as e:
            sharded_ef = {"error": str(e)}
    info = rccl_info(dist, world, device) if dist is not None else None

    if rank == 0:
        res = {
            "metric": "IDs encoded+decoded / sec (ROC/ANS, bit-exact vs codec.cpp)" if args.codec == "roc"
            else f"IDs encoded+decoded / sec ({args.codec})",
            "value": None if dry else world * ntotal * args.steps / elapsed,
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
                         "class_streams": ctx.class_streams() if ctx is not None else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_over_algorithmic": (traffic / ((alg_per_id + perm_per_id) * ntotal)) if traffic else None,
                         "algorithmic_bytes_per_id": alg_per_id, "algorithmic_bytes_per_step": alg_bytes,
                         # the container path's extra 4 B/id (sampling permutation) counted too
                         "frac_with_perm": achieved_perm / HBM_PEAK_GBS, "algorithmic_bytes_per_id_with_perm": alg_per_id + perm_per_id,
                         "formula": "(16 + 2c) B/id x ids / (encode + compaction + decode kernel ms, hipEvents); traffic = PMC "
                                    "FETCH_SIZE + WRITE_SIZE per step (profiles/pmc_traffic_s1.json, separate passes) vs the bytes incl. perm"},
        }
        if dry:
            res["dry_run"] = True
        if info is not None:
            res["rccl"] = info
        full = None  # the unabridged secondary measurements (detail file)
        if dominant is not None:
            res["roofline"]["dominant_kernel"] = dominant
        if world == 1 and not args.no_extra and args.codec == "roc" and args.workload == "s1" and not dry:
            # other regimes of the same kernels, for context only (never part of `value`): many equal lists
            # (no long serial chain) and the two bandwidth-bound codecs of the same plugin surface
            full = {}
            try:
                if not args.no_s2:  # BASELINE north_star's roofline workload: 1 B ids on one GPU, through the three codecs
                    torch.cuda.empty_cache()
                    ws2 = synth.workload("s2", seed=1042 + rank)
                    full["s2"] = secondary(ws2, "roc", steps=3, floor=True, traffic_tag="s2_roc")
                    full["s2_elias_fano"] = secondary(ws2, "ef", steps=3, traffic_tag="s2_ef")
                    full["s2_packed_bits"] = secondary(ws2, "packed", steps=3, traffic_tag="s2_packed")
                    del ws2
                    ctx.trim()
                    torch.cuda.empty_cache()
                full["roc_many_equal_lists"] = secondary("uniform_16m", "roc", traffic_tag="uniform_16m_roc")
                full["packed_bits"] = secondary("uniform_16m", "packed")
                full["elias_fano"] = secondary("uniform_16m", "ef")
                # the size a search issues (S1: 1 M ids / 1024 lists): launch-bound, a few kernels of ~10 us each
                full["s1_elias_fano"] = secondary(wl, "ef")
                full["s1_packed_bits"] = secondary(wl, "packed")
                full["c5"] = secondary("c5", "roc", floor=True)
                full["graph_rows"] = secondary_graph()
                full["wavelet_tree"] = secondary_wavelet()
                res["extra"] = {k: compact(v) for k, v in full.items()}
                g = full["graph_rows"]
                res["extra"]["graph_rows"] = {"workload": g["workload"], **{
                    n: {"edges_per_s": g[n]["edges_per_s"], "ms": g[n]["ms_per_step"],
                        "k_ms": [g[n]["kernel_ms"]["encode"], g[n]["kernel_ms"]["decode"]], "host_ms": g[n]["host_ms_per_step"],
                        "bits": g[n]["bits_per_edge"], "frac": g[n]["frac"], "frac_wall": g[n]["frac_wall"],
                        "ok": g[n]["roundtrip_ok"]} for n in ("roc", "elias_fano", "compact")}}
                wv = full["wavelet_tree"]
                res["extra"]["wavelet_tree"] = {"workload": wv["workload"], **{
                    k: {"build_ms": wv[k]["build_ms"], "decode_all_ms": wv[k]["decode_all_ms"],
                        "k_ms": [wv[k]["kernel_ms"]["build"], wv[k]["kernel_ms"]["decode_all"]], "selects_per_s": wv[k]["selects_per_s"],
                        "bits": wv[k]["bits_per_id"], "frac": wv[k]["frac"], "ok": wv[k]["ok"]} for k in ("wt_type_0", "wt_type_1")}}
                res["extra_legend"] = EXTRA_LEGEND
            except Exception as e:
                res["extra"] = {"error": str(e), **{k: compact(v) for k, v in full.items()}}
        for key, shd in (("sharded_c5", sharded), ("sharded_c5_ef", sharded_ef)):
            if shd is not None:
                res.setdefault("extra", {})[key] = (shd if "error" in shd else {
                    "workload": shd["config"]["workload"], "codec": shd["config"]["codec"], "ids_per_s": shd["value"], "ms": shd["ms_per_step"],
                    "steps": shd["steps"], "per_rank": shd["per_rank"], "gather_bytes": shd["gather_bytes"],
                    "gather_lists": shd["gather_lists"], "gather_verified": shd["gather_verified"], "scaling": "strong",
                    "strong_bound": shd["strong_bound"]})
        if world == 1 and args.codec == "roc" and not args.no_extra and not dry:
            try:
                cf = chain_floor(wl, d_ids)
                cf["chain_floor_ms"] = cf["encode_ms"] + cf["decode_ms"]
                cf["share_of_kernel_time"] = cf["chain_floor_ms"] / (1e3 * kern_s)
                res["roofline"]["chain_floor"] = cf
                if args.workload == "s1":
                    res["roofline"]["issue"] = committed_issue("s1_roc", 1e3 * kern_s)
                res["roofline"]["note"] = "latency bound: the batch cannot finish before its longest list's chain of dependent codec steps (chain_floor)"
            except Exception as e:
                res["roofline"]["chain_floor"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and args.codec == "roc" and ids_host is not None and not dry:
            try:
                res["cpu_baseline"] = cpu_baseline(offsets, ids_host)
            except Exception as e:  # the checker libraries are optional at bench time
                res["cpu_baseline"] = {"value": None, "unit": "IDs/s (encode+decode)", "cores": 0, "kind": "port",
                                       "sample": f"unavailable: {e}"}
        if full:
            detail = dict(res)
            detail["extra_full"] = full
            res["detail"] = write_detail(detail)
        print(json.dumps(sig(res)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
