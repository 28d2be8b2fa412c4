"""ctypes binding of libbiogpu.so (include/biogpu.h).  No CPU fallback: if the library or a
gfx950 device is missing, everything that needs the GPU raises."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libbiogpu.so")
CSRC = os.path.join(_HERE, "csrc")

BG_OK = 0
ERRORS = {-1: "INVALID_ARG", -2: "NO_DEVICE", -3: "HIP", -4: "OOM", -5: "SENTINEL",
          -6: "POSITIVE_PENALTY", -7: "OUT_OF_ALPHABET", -8: "TOO_LARGE", -9: "OPS_CAP",
          -10: "TRACEBACK", -11: "UNSUPPORTED", -12: "IO"}
MIN_SCORE = -858993459


class BiogpuError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib().bg_strerror(status).decode()
        last = lib().bg_last_error().decode()
        super().__init__(f"{where}: {ERRORS.get(status, status)}: {msg}" + (f" [{last}]" if last else ""))


class SentinelError(BiogpuError, ValueError):
    """suffix_array.rs:431-437 assert"""


class PenaltyError(BiogpuError, AssertionError):
    """pairwise/mod.rs:265-266,554-571 asserts"""


class AlphabetError(BiogpuError, IndexError):
    """fmindex.rs:229 / bwt.rs:158 index out of bounds"""


_EXC = {-5: SentinelError, -6: PenaltyError, -7: AlphabetError}


def check(status, where=""):
    if status != BG_OK:
        raise _EXC.get(status, BiogpuError)(status, where)


class ScoringC(C.Structure):
    _fields_ = [("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("xclip_prefix", C.c_int32), ("xclip_suffix", C.c_int32),
                ("yclip_prefix", C.c_int32), ("yclip_suffix", C.c_int32),
                ("match_score", C.c_int32), ("mismatch_score", C.c_int32),
                ("match_scores_some", C.c_int32),
                ("matrix", C.POINTER(C.c_int32))]


class TimingC(C.Structure):
    _fields_ = [("fill_ms", C.c_float), ("traceback_ms", C.c_float), ("fm_ms", C.c_float),
                ("fill_launches", C.c_uint32), ("traceback_launches", C.c_uint32),
                ("fm_launches", C.c_uint32)]


# bg_alignment_t
ALN_DTYPE = np.dtype([("score", "<i4"), ("xstart", "<u4"), ("xend", "<u4"), ("ystart", "<u4"),
                      ("yend", "<u4"), ("xlen", "<u4"), ("ylen", "<u4"), ("n_ops", "<u4"),
                      ("ops_off", "<u8"), ("clip_len", "<u4", (4,)), ("n_clips", "u1"),
                      ("mode", "u1"), ("status", "i1"), ("_pad", "u1"), ("_tail", "<u4")])
assert ALN_DTYPE.itemsize == 64, ALN_DTYPE.itemsize
# bg_seed_hit_t
SEED_HIT_DTYPE = np.dtype([("aln", ALN_DTYPE), ("window_start", "<u8"), ("ref_start", "<u8"), ("ref_end", "<u8"),
                           ("n_candidates", "<u4"), ("n_seed_hits", "<u4")])
assert SEED_HIT_DTYPE.itemsize == 96, SEED_HIT_DTYPE.itemsize


class SeedParamsC(C.Structure):
    _fields_ = [("seed_len", C.c_uint32), ("stride", C.c_uint32), ("max_occ", C.c_uint32), ("pad", C.c_uint32)]


# bg_fastq_record_t
FQREC_DTYPE = np.dtype([("id_off", "<u8"), ("desc_off", "<u8"), ("seq_off", "<u8"), ("qual_off", "<u8"),
                        ("id_len", "<u4"), ("desc_len", "<u4"), ("seq_len", "<u4"), ("qual_len", "<u4"),
                        ("has_desc", "<i4"), ("check", "<i4")])
assert FQREC_DTYPE.itemsize == 56, FQREC_DTYPE.itemsize

SYMBOLS = ["bg_device_count", "bg_init", "bg_free", "bg_strerror", "bg_last_error",
           "bg_set_option", "bg_suffix_array", "bg_bwt", "bg_less", "bg_fm_build", "bg_fm_free",
           "bg_fm_device_bytes", "bg_fm_set_option", "bg_fm_backward_search_batch", "bg_fm_backward_search_batch_dev",
           "bg_fm_set_suffix_array", "bg_fm_set_sampled_suffix_array", "bg_sa_get_batch", "bg_sa_get_batch_dev",
           "bg_interval_occ_batch", "bg_interval_occ_batch_dev", "bg_fmd_smems_batch", "bg_fmd_smems_batch_dev", "bg_fmd_interval_batch",
           "bg_align_batch", "bg_align_batch_dev", "bg_align_banded_batch", "bg_align_banded_batch_dev", "bg_band_create_batch",
           "bg_align_banded_bands_batch", "bg_band_from_matches_batch", "bg_sparse_find_kmer_matches", "bg_sparse_sdpkpp",
           "bg_sparse_lcskpp", "bg_sparse_sdpkpp_union_lcskpp_path", "bg_sparse_expand_kmer_matches", "bg_fastq_parse",
           "bg_fastq_parse_dev", "bg_cigar_batch", "bg_cigar_batch_dev", "bg_get_timing", "bg_enable_timing", "bg_band_redo_pairs", "bg_pack2_host",
           "bg_pretty_batch", "bg_suffix_array_dev", "bg_bwt_dev", "bg_sa_sample_dev", "bg_suffix_array_dev64", "bg_bwt_dev64", "bg_sa_sample_dev64", "bg_fm_build_dev", "bg_fm_set_text", "bg_fm_set_text_dev", "bg_seed_extend_batch", "bg_seed_extend_batch_dev",
           "bg_pack2_dev", "bg_unpack2_dev", "bg_fm_pattern_codes", "bg_fm_backward_search_packed_dev",
           "bg_fm_backward_search_count_lines_dev", "bg_align_batch_packed_dev", "bg_fm_step2_bytes",
           "bg_shard_range", "bg_shard_balanced", "bg_comm_unique_id", "bg_comm_init", "bg_comm_init_host",
           "bg_gather_records", "bg_gather_records_cap", "bg_gather_records_host", "bg_comm_free", "bg_fm_save", "bg_fm_load",
           "bg_fm_len", "bg_fm_less", "bg_fm_bwt", "bg_fm_bwt_dev", "bg_comm_world",
           "bg_fmd_smems_batch64", "bg_fmd_smems_batch64_dev", "bg_fmd_interval_batch64"]


def build(force=False):
    """Compile every HIP source for gfx950 into rust-bio_amd/libbiogpu.so (hipcc
    cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (the engine has no CPU fallback)")
        L = C.CDLL(SO_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.bg_strerror.restype = C.c_char_p
        L.bg_strerror.argtypes = [i32]
        L.bg_last_error.restype = C.c_char_p
        L.bg_device_count.restype = i32
        L.bg_init.argtypes = [i32, C.POINTER(vp)]
        L.bg_free.argtypes = [vp]
        L.bg_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.bg_suffix_array.argtypes = [vp, u64, vp]
        L.bg_bwt.argtypes = [vp, vp, u64, vp]
        L.bg_less.argtypes = [vp, u64, vp, u32, vp, C.POINTER(u32)]
        L.bg_fm_build.argtypes = [vp, vp, u64, vp, u32, u32, vp, u32, C.POINTER(vp)]
        L.bg_fm_free.argtypes = [vp]
        L.bg_fm_save.argtypes = [vp, C.c_char_p]
        L.bg_fm_load.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
        L.bg_fm_len.argtypes = [vp, C.POINTER(u64)]
        L.bg_fm_less.argtypes = [vp, vp, C.POINTER(u32)]
        L.bg_fm_bwt.argtypes = [vp, vp]
        L.bg_fm_bwt_dev.argtypes = [vp, vp, vp]
        L.bg_fm_build_dev.argtypes = [vp, vp, u64, u32, vp, u32, vp, C.POINTER(vp), vp]
        L.bg_fm_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.bg_fm_device_bytes.restype = u64
        L.bg_fm_device_bytes.argtypes = [vp]
        L.bg_fm_step2_bytes.restype = u64
        L.bg_fm_step2_bytes.argtypes = [vp]
        L.bg_shard_range.argtypes = [u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
        L.bg_shard_balanced.argtypes = [vp, u64, i32, vp]
        L.bg_comm_unique_id.argtypes = [vp]
        L.bg_comm_init.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
        L.bg_comm_init_host.argtypes = [vp, i32, i32, C.c_char_p, C.POINTER(vp)]
        L.bg_gather_records.argtypes = [vp, vp, u64, u32, vp, vp, vp]
        L.bg_gather_records_cap.argtypes = [vp, vp, u64, u32, vp, u64, vp, vp]
        L.bg_gather_records_host.argtypes = [vp, vp, u64, u32, vp, u64, vp]
        L.bg_comm_free.argtypes = [vp]
        L.bg_comm_world.argtypes = [vp, vp]
        L.bg_fm_backward_search_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
        L.bg_fm_backward_search_batch_dev.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp]
        L.bg_fmd_interval_batch.argtypes = [vp, u64, vp, vp, vp, vp]
        L.bg_fmd_smems_batch.argtypes = [vp, i32, u64, vp, vp, vp, u32, u32, vp, vp]
        L.bg_fmd_smems_batch_dev.argtypes = [vp, i32, u64, vp, vp, vp, u32, u32, u32, vp, vp, vp]
        L.bg_fmd_interval_batch64.argtypes = [vp, u64, vp, vp, vp, vp]
        L.bg_fmd_smems_batch64.argtypes = [vp, i32, u64, vp, vp, vp, u32, u32, vp, vp]
        L.bg_fmd_smems_batch64_dev.argtypes = [vp, i32, u64, vp, vp, vp, u32, u32, u32, vp, vp, vp]
        L.bg_fm_set_suffix_array.argtypes = [vp, vp, u64]
        L.bg_fm_set_sampled_suffix_array.argtypes = [vp, vp, u64, u32, C.c_uint8, vp, vp, u64]
        L.bg_sa_get_batch.argtypes = [vp, u64, vp, vp]
        L.bg_fastq_parse.argtypes = [vp, vp, u64, vp, u64, vp, vp, vp, vp, C.POINTER(u64), C.POINTER(i32), C.POINTER(u64)]
        L.bg_fastq_parse_dev.argtypes = [vp, vp, u64, vp, u64, vp, vp, vp, vp, C.POINTER(u64), C.POINTER(i32), C.POINTER(u64), vp]
        L.bg_cigar_batch.argtypes = [vp, u64, vp, vp, u64, i32, vp, u64, vp]
        L.bg_cigar_batch_dev.argtypes = [vp, u64, vp, vp, i32, vp, u64, vp, vp]
        L.bg_sa_get_batch_dev.argtypes = [vp, u64, vp, vp, vp]
        L.bg_interval_occ_batch.argtypes = [vp, u64, vp, vp, vp, vp, u64]
        L.bg_interval_occ_batch_dev.argtypes = [vp, u64, vp, vp, u64, vp, vp]
        L.bg_align_batch.argtypes = [vp, C.POINTER(ScoringC), i32, u64, vp, vp, vp, vp, vp, vp,
                                     u64, C.POINTER(u64)]
        L.bg_align_batch_dev.argtypes = [vp, C.POINTER(ScoringC), i32, u64, vp, vp, vp, vp, u32,
                                         u32, vp, vp, u64, vp]
        L.bg_align_banded_batch.argtypes = [vp, C.POINTER(ScoringC), i32, u32, u32, u64, vp, vp,
                                            vp, vp, vp, vp, u64, C.POINTER(u64), vp]
        L.bg_align_banded_batch_dev.argtypes = [vp, C.POINTER(ScoringC), i32, u32, u32, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]
        L.bg_band_create_batch.argtypes = [C.POINTER(ScoringC), i32, u32, u32, u64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.bg_align_banded_bands_batch.argtypes = [vp, C.POINTER(ScoringC), i32, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                  u64, C.POINTER(u64), vp]
        L.bg_band_from_matches_batch.argtypes = [C.POINTER(ScoringC), i32, u32, u32, u64, vp, vp, vp, vp, vp, vp, vp, vp,
                                                 vp, vp]
        for name, args in [("bg_sparse_find_kmer_matches", [vp, u64, vp, u64, u32, vp, u64]),
                           ("bg_sparse_sdpkpp", [vp, u64, u32, u32, C.c_int32, C.c_int32, vp, u64]),
                           ("bg_sparse_lcskpp", [vp, u64, u32, vp, u64, C.POINTER(u32)]),
                           ("bg_sparse_sdpkpp_union_lcskpp_path", [vp, u64, u32, u32, C.c_int32, C.c_int32, vp, u64]),
                           ("bg_sparse_expand_kmer_matches", [vp, u64, vp, u64, u32, vp, u64, u32, vp, u64])]:
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = args
        L.bg_pretty_batch.argtypes = [vp, u64, vp, vp, u64, vp, vp, vp, vp, u32, vp, u64, vp]
        L.bg_suffix_array_dev.argtypes = [vp, vp, u64, vp, vp]
        L.bg_bwt_dev.argtypes = [vp, vp, vp, u64, vp, vp]
        L.bg_sa_sample_dev.argtypes = [vp, vp, vp, u64, u32, C.c_uint8, vp, vp, vp, u64, C.POINTER(u64), vp]
        L.bg_suffix_array_dev64.argtypes = [vp, vp, u64, vp, vp]
        L.bg_bwt_dev64.argtypes = [vp, vp, vp, u64, vp, vp]
        L.bg_sa_sample_dev64.argtypes = [vp, vp, vp, u64, u32, C.c_uint8, vp, vp, vp, u64, C.POINTER(u64), vp]
        L.bg_fm_set_text.argtypes = [vp, vp, u64]
        L.bg_fm_set_text_dev.argtypes = [vp, vp, u64]
        L.bg_seed_extend_batch.argtypes = [vp, C.POINTER(ScoringC), C.POINTER(SeedParamsC), u64, vp, vp, vp, vp, u64, C.POINTER(u64)]
        L.bg_seed_extend_batch_dev.argtypes = [vp, C.POINTER(ScoringC), C.POINTER(SeedParamsC), u64, vp, vp, u32, vp, vp, u64, vp, vp]
        L.bg_pack2_dev.argtypes = [vp, vp, u64, vp, vp, vp, vp]
        L.bg_unpack2_dev.argtypes = [vp, vp, u64, vp, vp, vp]
        L.bg_fm_pattern_codes.argtypes = [vp, vp]
        L.bg_fm_backward_search_packed_dev.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp]
        L.bg_fm_backward_search_count_lines_dev.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, C.POINTER(u64), vp]
        L.bg_align_batch_packed_dev.argtypes = [vp, C.POINTER(ScoringC), i32, u64, vp, vp, vp, vp, vp, u32, u32, vp, vp, u64, vp]
        L.bg_get_timing.argtypes = [vp, C.POINTER(TimingC)]
        L.bg_enable_timing.argtypes = [vp, i32]
        L.bg_band_redo_pairs.argtypes = [vp, C.POINTER(u64)]
        L.bg_pack2_host.argtypes = [vp, u64, vp, vp]
        for s in SYMBOLS:
            if getattr(L, s).restype is C.c_int or s.startswith("bg_") and getattr(L, s).restype is None:
                pass
        _lib = L
    return _lib


class Context:
    """bg_ctx: one per device, not thread-safe (like `&mut Aligner`)."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        check(lib().bg_init(device, C.byref(self.h)), "bg_init")
        self.device = device

    def set_option(self, key, value):
        check(lib().bg_set_option(self.h, key.encode(), int(value)), "bg_set_option")

    def enable_timing(self, on=True):
        check(lib().bg_enable_timing(self.h, 1 if on else 0))

    def timing(self):
        t = TimingC()
        check(lib().bg_get_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in TimingC._fields_}

    def band_redo_pairs(self):
        """pairs of the last banded call that the packed fill flagged and the int32 kernels recomputed"""
        v = C.c_uint64(0)
        check(lib().bg_band_redo_pairs(self.h, C.byref(v)))
        return int(v.value)

    def close(self):
        if self.h:
            lib().bg_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if lib().bg_device_count() > 1 else 0
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def as_u8(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def concat(seqs):
    """list of bytes -> (uint8 buffer, uint64 offsets[n+1])"""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8)
    return buf, off
