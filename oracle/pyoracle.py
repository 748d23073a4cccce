"""ctypes access to the test-only checkers in oracle/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

  Oracle  -> oracle/_build/libvidc_oracle.so  (clean-room C restatement, vidc_oracle.c)
  Ref     -> oracle/_ref/libvidc_ref.so       (the reference's own codec.cpp + our driver)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libvidc_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvidc_ref.so")

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """(Re)build the checkers with oracle/Makefile (reference part only if /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.exists("/root/reference/custom_invlist_cpp/codec.cpp") and not os.path.exists(REF_SO)
    ):
        subprocess.check_call(["make", "-C", HERE, "-s"])


class _AnsState(C.Structure):
    _fields_ = [
        ("head", C.c_uint64),
        ("stack", C.POINTER(C.c_uint32)),
        ("nstack", C.c_size_t),
        ("cap", C.c_size_t),
        ("mt_draws", C.c_uint32),
    ]


class _Ef(C.Structure):
    _fields_ = [
        ("universe", C.c_uint64),
        ("m", C.c_uint64),
        ("l", C.c_int),
        ("low_nbits", C.c_uint64),
        ("high_nbits", C.c_uint64),
        ("low", C.POINTER(C.c_uint64)),
        ("high", C.POINTER(C.c_uint64)),
    ]


class Oracle:
    """Clean-room CPU restatement (vidc_oracle.c)."""

    def __init__(self):
        build()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.vo_mt19937_table.argtypes = [C.c_uint32, _u32p, C.c_size_t]
        L.vo_precision_from_max_id.argtypes = [C.c_int32]
        L.vo_precision_from_max_id.restype = C.c_int
        L.vo_ans_init.argtypes = [C.POINTER(_AnsState)]
        L.vo_ans_free.argtypes = [C.POINTER(_AnsState)]
        L.vo_roc_encode.argtypes = [C.c_size_t, _u64p, C.c_int, C.POINTER(_AnsState), _u64p, _u32p]
        L.vo_roc_decode.argtypes = [C.POINTER(_AnsState), C.c_size_t, C.c_int, _u64p]
        L.vo_packed_bits_for.argtypes = [C.c_uint64]
        L.vo_packed_bits_for.restype = C.c_int
        L.vo_packed_write.argtypes = [_u8p, C.c_size_t, C.c_uint64, C.c_int]
        L.vo_packed_read.argtypes = [_u8p, C.c_size_t, C.c_int]
        L.vo_packed_read.restype = C.c_uint64
        L.vo_ef_low_bits.argtypes = [C.c_uint64, C.c_uint64]
        L.vo_ef_low_bits.restype = C.c_int
        L.vo_ef_build.argtypes = [C.POINTER(_Ef), C.c_uint64, C.c_uint64, _u64p]
        L.vo_ef_free.argtypes = [C.POINTER(_Ef)]
        L.vo_ef_select.argtypes = [C.POINTER(_Ef), C.c_uint64]
        L.vo_ef_select.restype = C.c_uint64
        L.vo_ef_decode_all.argtypes = [C.POINTER(_Ef), _u64p]
        L.vo_wt_select.argtypes = [_u32p, C.c_size_t, C.c_uint32, C.c_uint64]
        L.vo_wt_select.restype = C.c_int64
        L.vo_roc_bench_lists.argtypes = [C.c_size_t, _u64p, _u64p, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.vo_roc_bench_lists.restype = C.c_size_t
        L.vo_omp_max_threads.restype = C.c_int

    # -- helpers
    def mt_table(self, count, seed=1234):
        out = np.zeros(count, dtype=np.uint32)
        self.lib.vo_mt19937_table(seed, out, count)
        return out

    def precision(self, max_id):
        return int(self.lib.vo_precision_from_max_id(int(np.int32(np.uint64(max_id) & np.uint64(0xFFFFFFFF)))))

    def list_precision(self, ids):
        """custom_invlists_impl.cpp:163-164 on one list (ids as uint64)."""
        ids = np.asarray(ids, dtype=np.uint64)
        if ids.size == 0:
            return 0
        return self.precision(int(ids.max()))

    # -- ROC
    def roc_encode(self, ids, precision):
        """-> dict(head, words, order, perm, mt_draws)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        n = ids.size
        st = _AnsState()
        self.lib.vo_ans_init(C.byref(st))
        order = np.zeros(max(n, 1), dtype=np.uint64)
        perm = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.vo_roc_encode(n, ids if n else np.zeros(1, np.uint64), precision, C.byref(st), order, perm)
        words = np.ctypeslib.as_array(st.stack, shape=(st.nstack,)).copy() if st.nstack else np.zeros(0, np.uint32)
        res = dict(head=int(st.head), words=words.astype(np.uint32), order=order[:n], perm=perm[:n],
                   mt_draws=int(st.mt_draws))
        self.lib.vo_ans_free(C.byref(st))
        return res

    def roc_decode(self, head, words, n, precision, mt_draws=0):
        """-> (decoded ids in sampling order, end_head, end_words, end_mt_draws)."""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        st = _AnsState()
        buf = (C.c_uint32 * (len(words) + 16))()
        # the oracle reallocs its stack: give it a malloc'ed copy through vo_ans_copy semantics
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        cap = len(words) + 16
        ptr = libc.malloc(cap * 4)
        C.memmove(ptr, words.ctypes.data, len(words) * 4)
        st.head = head
        st.stack = C.cast(ptr, C.POINTER(C.c_uint32))
        st.nstack = len(words)
        st.cap = cap
        st.mt_draws = mt_draws
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self.lib.vo_roc_decode(C.byref(st), n, precision, out)
        end_words = np.ctypeslib.as_array(st.stack, shape=(st.nstack,)).copy() if st.nstack else np.zeros(0, np.uint32)
        res = (out[:n], int(st.head), end_words.astype(np.uint32), int(st.mt_draws))
        self.lib.vo_ans_free(C.byref(st))
        del buf
        return res

    # -- packed bits
    def packed_bits_for(self, ntotal):
        return int(self.lib.vo_packed_bits_for(ntotal))

    def packed_encode(self, ids, bits):
        ids = np.asarray(ids, dtype=np.uint64)
        nbytes = (ids.size * bits + 7) // 8
        code = np.zeros(max(nbytes, 1), dtype=np.uint8)
        for i, x in enumerate(ids):
            self.lib.vo_packed_write(code, i * bits, int(x), bits)
        return code[:nbytes]

    def packed_decode(self, code, n, bits):
        code = np.ascontiguousarray(np.concatenate([np.asarray(code, np.uint8), np.zeros(8, np.uint8)]))
        return np.array([self.lib.vo_packed_read(code, i * bits, bits) for i in range(n)], dtype=np.uint64)

    # -- Elias-Fano
    def ef_build(self, sorted_ids, universe=None):
        """-> dict(l, low_nbits, high_nbits, low(words u64), high(words u64), decoded)."""
        ids = np.ascontiguousarray(sorted_ids, dtype=np.uint64)
        m = ids.size
        if universe is None:
            universe = int(ids.max()) if m else 0
        ef = _Ef()
        self.lib.vo_ef_build(C.byref(ef), universe, m, ids if m else np.zeros(1, np.uint64))
        nlow = (ef.low_nbits + 63) // 64
        nhigh = (ef.high_nbits + 63) // 64
        low = np.ctypeslib.as_array(ef.low, shape=(nlow + 1,))[:nlow].copy()
        high = np.ctypeslib.as_array(ef.high, shape=(nhigh + 1,))[:nhigh].copy()
        dec = np.zeros(max(m, 1), dtype=np.uint64)
        self.lib.vo_ef_decode_all(C.byref(ef), dec)
        sel = np.array([self.lib.vo_ef_select(C.byref(ef), i) for i in range(min(m, 64))], dtype=np.uint64)
        res = dict(l=int(ef.l), low_nbits=int(ef.low_nbits), high_nbits=int(ef.high_nbits), low=low, high=high,
                   decoded=dec[:m], select_head=sel)
        self.lib.vo_ef_free(C.byref(ef))
        return res

    def wt_select(self, list_nos, c, k):
        list_nos = np.ascontiguousarray(list_nos, dtype=np.uint32)
        return int(self.lib.vo_wt_select(list_nos, list_nos.size, c, k))

    # -- CPU baseline ("port")
    def max_threads(self):
        return int(self.lib.vo_omp_max_threads())

    def bench_lists(self, offsets, ids, threads):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        te, td, nb = C.c_double(), C.c_double(), C.c_uint64()
        bad = self.lib.vo_roc_bench_lists(offsets.size - 1, offsets, ids, threads, C.byref(te), C.byref(td), C.byref(nb))
        return dict(t_enc=te.value, t_dec=td.value, bytes=int(nb.value), bad_lists=int(bad))


class Ref:
    """The reference's own codec.cpp, compiled by oracle/Makefile into oracle/_ref/."""

    @staticmethod
    def available():
        build()
        return os.path.exists(REF_SO)

    def __init__(self):
        build()
        L = self.lib = C.CDLL(REF_SO)
        szp = C.POINTER(C.c_size_t)
        u64pp = C.POINTER(C.c_uint64)
        L.rc_compress.argtypes = [C.c_size_t, _u64p, C.c_int, u64pp, _u32p, C.c_size_t, szp]
        L.rc_container_encode.argtypes = [C.c_size_t, _u64p, C.c_int, C.c_uint32, u64pp, _u32p, C.c_size_t, szp,
                                          _u64p, _u32p]
        L.rc_decompress.argtypes = [C.c_uint64, _u32p, C.c_size_t, C.c_size_t, C.c_int, _u64p, u64pp, _u32p,
                                    C.c_size_t, szp]
        L.rc_roundtrip.argtypes = [C.c_size_t, _u64p, C.c_int, u64pp, _u32p, C.c_size_t, szp, _u64p]
        L.rc_bench_lists.argtypes = [C.c_size_t, _u64p, _u64p, C.c_int,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.rc_bench_lists.restype = C.c_size_t
        L.rc_omp_max_threads.restype = C.c_int

    @staticmethod
    def _cap(n, precision):
        return n * (max(precision, 0) + 8) // 32 + 64

    def compress(self, ids, precision):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        cap = self._cap(ids.size, precision)
        words = np.zeros(cap, dtype=np.uint32)
        head, nw = C.c_uint64(), C.c_size_t()
        rc = self.lib.rc_compress(ids.size, ids, precision, C.byref(head), words, cap, C.byref(nw))
        assert rc == 0
        return dict(head=int(head.value), words=words[: nw.value].copy())

    def container_encode(self, ids, precision, shuffle_seed=7):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        n = ids.size
        cap = self._cap(n, precision)
        words = np.zeros(cap, dtype=np.uint32)
        order = np.zeros(max(n, 1), dtype=np.uint64)
        perm = np.zeros(max(n, 1), dtype=np.uint32)
        head, nw = C.c_uint64(), C.c_size_t()
        rc = self.lib.rc_container_encode(n, ids, precision, shuffle_seed, C.byref(head), words, cap, C.byref(nw),
                                          order, perm)
        assert rc == 0
        return dict(head=int(head.value), words=words[: nw.value].copy(), order=order[:n], perm=perm[:n])

    def decompress(self, head, words, n, precision):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        cap = words.size + 64
        ew = np.zeros(cap, dtype=np.uint32)
        eh, enw = C.c_uint64(), C.c_size_t()
        rc = self.lib.rc_decompress(head, words if words.size else np.zeros(1, np.uint32), words.size, n, precision,
                                    out, C.byref(eh), ew, cap, C.byref(enw))
        assert rc == 0
        return out[:n], int(eh.value), ew[: enw.value].copy()

    def roundtrip(self, ids, precision):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        n = ids.size
        cap = self._cap(n, precision)
        words = np.zeros(cap, dtype=np.uint32)
        dec = np.zeros(max(n, 1), dtype=np.uint64)
        head, nw = C.c_uint64(), C.c_size_t()
        rc = self.lib.rc_roundtrip(n, ids, precision, C.byref(head), words, cap, C.byref(nw), dec)
        assert rc == 0
        return dict(head=int(head.value), words=words[: nw.value].copy(), decoded=dec[:n])

    def max_threads(self):
        return int(self.lib.rc_omp_max_threads())

    def bench_lists(self, offsets, ids, threads):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        te, td, nb = C.c_double(), C.c_double(), C.c_uint64()
        bad = self.lib.rc_bench_lists(offsets.size - 1, offsets, ids, threads, C.byref(te), C.byref(td), C.byref(nb))
        return dict(t_enc=te.value, t_dec=td.value, bytes=int(nb.value), bad_lists=int(bad))
