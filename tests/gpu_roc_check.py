"""Ad-hoc GPU parity driver (used through gpurun while developing): HIP ROC vs golden + oracle."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_cases import CASES, make_ids, fnv_stream, fnv_u64
from oracle.pyoracle import Oracle
from vector_db_id_compression_amd.codecs import RocLists

def main():
    o = Oracle()
    gold = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests/golden/roc_golden.json")))["cases"]}
    nbad = 0
    names = [c["name"] for c in CASES]
    lists = [make_ids(c) for c in CASES]
    t0 = time.time()
    for name, ids, case in zip(names, lists, CASES):
        g = gold[name]
        off = np.array([0, ids.size], dtype=np.uint64)
        mode = case.get("precision", -1)
        try:
            r = RocLists.encode(off, ids, precision_mode=mode if mode is not None else -1, want_perm=True)
            info = r.info()
            words = r.words(0)
            ok_enc = int(info["heads"][0]) == g["head"] and int(info["nwords"][0]) == g["nwords"] and \
                fnv_stream(int(info["heads"][0]), words) == g["stream_fnv"] and int(info["precision"][0]) == g["precision"]
            perm = r.perm()
            ok_perm = fnv_u64(perm.astype(np.uint64)) == g["perm_fnv"]
            dec = r.decode_all().cpu().numpy().view(np.uint64)
            ok_dec = fnv_u64(dec) == g["decoded_fnv"]
            clean = r.last_decode_nonclean == 0
            flag = "OK " if (ok_enc and ok_perm and ok_dec) else "BAD"
            if flag == "BAD": nbad += 1
            print(f"{flag} {name:36s} n={ids.size:6d} P={g['precision']:2d} enc={ok_enc} perm={ok_perm} dec={ok_dec} clean={clean} "
                  f"head={int(info['heads'][0])} nw={int(info['nwords'][0])}/{g['nwords']} draws={int(info['mt_draws'][0])}", flush=True)
            if not ok_enc and ids.size <= 96:
                print("   ids", ids.tolist(), "\n   got words", words.tolist(), "want", g.get("words"), "want head", g["head"])
            if ok_enc and not ok_dec and ids.size <= 96:
                print("   got dec", dec.tolist(), "want", g.get("decoded"))
        except Exception as e:
            nbad += 1
            print(f"EXC {name}: {e}", flush=True)
    print("golden cases bad:", nbad, "time", time.time() - t0)
    # multi-list batch against the oracle
    rng = np.random.default_rng(1)
    sizes = np.concatenate([rng.integers(0, 70, 200), rng.integers(60, 3000, 60), [5000, 9000, 0, 1, 2, 40000]])
    rng.shuffle(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    allids = []
    for s in sizes:
        allids.append(np.sort(rng.choice(1 << 20, size=int(s), replace=False)).astype(np.uint64))
    ids = np.concatenate(allids) if allids else np.zeros(0, np.uint64)
    t0 = time.time()
    r = RocLists.encode(off, ids, want_perm=True)
    torch.cuda.synchronize(); t1 = time.time()
    dec = r.decode_all().cpu().numpy().view(np.uint64); t2 = time.time()
    info = r.info(); perm = r.perm()
    bad = 0
    for l, s in enumerate(sizes):
        li = allids[l]
        if s == 0: continue
        P = o.list_precision(li)
        e = o.roc_encode(li, P)
        w = r.words(l, int(info["nwords"][l]))
        okl = int(info["heads"][l]) == e["head"] and np.array_equal(w, e["words"]) and int(info["precision"][l]) == P
        okl = okl and np.array_equal(perm[int(off[l]):int(off[l+1])], e["perm"])
        okl = okl and np.array_equal(dec[int(off[l]):int(off[l+1])], e["order"])
        if not okl:
            bad += 1
            if bad < 5: print("  batch mismatch list", l, "n", s)
    print(f"batch: {len(sizes)} lists {ids.size} ids bad={bad} enc {t1-t0:.3f}s dec {t2-t1:.3f}s bytes={r.compressed_bytes} nonclean={r.last_decode_nonclean}")
    # unsorted input + decode_lists
    ids2 = ids.copy()
    for l in range(0, len(sizes), 3):
        seg = ids2[int(off[l]):int(off[l+1])]; rng.shuffle(seg)
    r2 = RocLists.encode(off, ids2, want_perm=True)
    info2 = r2.info(); perm2 = r2.perm(); bad2 = 0
    for l, s in enumerate(sizes):
        if s == 0: continue
        li = ids2[int(off[l]):int(off[l+1])]
        e = o.roc_encode(li, o.list_precision(li))
        okl = int(info2["heads"][l]) == e["head"] and np.array_equal(r2.words(l, int(info2["nwords"][l])), e["words"]) and np.array_equal(perm2[int(off[l]):int(off[l+1])], e["perm"])
        if not okl: bad2 += 1
    sel = np.array([5, 0, len(sizes)-1, 17, 5], dtype=np.uint64)
    d, doff = r2.decode_lists(sel); d = d.cpu().numpy().view(np.uint64); bad3 = 0
    full = r2.decode_all().cpu().numpy().view(np.uint64)
    for i, l in enumerate(sel):
        if not np.array_equal(d[int(doff[i]):int(doff[i+1])], full[int(off[int(l)]):int(off[int(l)+1])]): bad3 += 1
    print(f"unsorted batch bad={bad2} decode_lists bad={bad3}")
    # graph rows
    N, K = 500, 64
    rows = np.full((N, K), -1, dtype=np.int32)
    for i in range(N):
        d_ = int(rng.integers(0, K + 1))
        rows[i, :d_] = rng.choice(100000, size=d_, replace=False)
    rg = RocLists.encode_rows(rows)
    ig = rg.info(); badg = 0
    outr, cnt = rg.decode_rows(np.arange(N)); outr = outr.cpu().numpy()
    for i in range(N):
        d_ = int((rows[i] >= 0).sum())
        if d_ == 0:
            if cnt[i] != 0: badg += 1
            continue
        li = rows[i, :d_].astype(np.uint64)
        e = o.roc_encode(li, o.list_precision(li))
        okl = int(ig["heads"][i]) == e["head"] and np.array_equal(rg.words(i, int(ig["nwords"][i])), e["words"]) and cnt[i] == d_ \
            and np.array_equal(outr[i, :d_].astype(np.uint64), e["order"]) and np.all(outr[i, d_:] == -1)
        if not okl: badg += 1
    print(f"graph rows bad={badg} bytes={rg.compressed_bytes}")
    sys.exit(1 if (nbad or bad or bad2 or bad3 or badg) else 0)

if __name__ == "__main__":
    main()
