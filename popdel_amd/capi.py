"""ctypes mirror of include/popdel_hip.h.

``load_hip()`` opens the product library (popdel_amd/lib/libpopdel_hip.so) and raises if it is missing: there is no
CPU fallback in the product path. Tests and the bench's CPU-baseline leg may open the oracle library through
``load_library(path)`` with an explicit path; both expose the same symbols.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# POPDEL_HIP_LIBRARY points the binding at another build of the product library (A/B measurements); never at the oracle:
# load_hip() checks the backend name.
HIP_LIBRARY = os.environ.get("POPDEL_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libpopdel_hip.so")

ABI_SYMBOLS = (
    "pd_ctx_create", "pd_ctx_destroy", "pd_segment_submit", "pd_segment_stage", "pd_segment_run", "pd_segment_unstage",
    "pd_segment_collect", "pd_calls_free", "pd_stage1_last", "pd_last_timing", "pd_last_error", "pd_backend_name",
    "pd_abi_version", "pd_shard_samples", "pd_ctx_set_shard", "pd_ctx_set_shard_rccl", "pd_rccl_unique_id",
    "pd_local_group_create", "pd_local_group_destroy", "pd_ctx_set_shard_local", "pd_log_table_view",
    "pd_ctx_keep_results_on_device", "pd_calls_merge", "pd_calls_fetch_columns", "pd_merged_free",
    "pd_walk_submit", "pd_walk_supported", "pd_walk_active", "pd_device_warmup", "pd_inflate_members", "pd_inflate_last_error",
    "pd_inflater_create", "pd_inflater_destroy", "pd_inflater_input", "pd_inflater_run", "pd_inflater_run_parsed", "pd_inflater_run_summarized", "pd_walk_source_counts", "pd_walk_active_ex", "pd_inflater_run_id", "pd_ctx_set_shard_lanes",
    "pd_inflater_last_error", "pd_ctx_stage1_view", "pd_last_stage_timing", "pd_host_copy_gbs", "pd_host_regions_offer", "pd_walk_entry_counts",
)
RCCL_ID_BYTES = 128


class PdReadgroup(C.Structure):
    _fields_ = [("median", C.c_uint32), ("read_length", C.c_uint32), ("stddev", C.c_double), ("min_prob", C.c_double),
                ("hist_offset", C.c_int32), ("hist_len", C.c_uint32), ("hist_values", C.POINTER(C.c_double)),
                ("lower_quantile_dist", C.c_uint32), ("upper_quantile_dist", C.c_uint32), ("max_load", C.c_uint32),
                ("min_init_del_len", C.c_uint32)]


class PdConfig(C.Structure):
    _fields_ = [("iterations", C.c_uint32), ("min_len", C.c_uint32), ("min_likelihood_ratio", C.c_double),
                ("min_sample_fraction", C.c_double), ("somatic", C.c_int32), ("window_wise_output", C.c_int32)]


class PdSegment(C.Structure):
    _fields_ = [("first_window_pos", C.c_uint32), ("n_windows", C.c_uint32), ("window_shift", C.c_uint32),
                ("n_readgroups", C.c_uint32), ("rg_offsets", C.POINTER(C.c_uint64)), ("pair_dev", C.POINTER(C.c_int32)),
                ("pair_start", C.POINTER(C.c_uint32)), ("pair_on", C.POINTER(C.c_uint32)),
                ("pair_off", C.POINTER(C.c_uint32))]


class PdCalls(C.Structure):
    _fields_ = [("n_calls", C.c_uint32), ("n_samples", C.c_uint32), ("initial_length", C.POINTER(C.c_uint32)),
                ("iterations", C.POINTER(C.c_uint32)), ("deletion_length", C.POINTER(C.c_uint32)),
                ("likelihood_ratio", C.POINTER(C.c_double)), ("frequency", C.POINTER(C.c_double)),
                ("window_position", C.POINTER(C.c_uint32)), ("position", C.POINTER(C.c_uint32)),
                ("end_position", C.POINTER(C.c_uint32)), ("filter", C.POINTER(C.c_uint8)), ("pl", C.POINTER(C.c_uint32)),
                ("lad", C.POINTER(C.c_uint32)), ("dad", C.POINTER(C.c_uint32)), ("first_last", C.POINTER(C.c_uint32)),
                ("near_integer_shifts", C.POINTER(C.c_uint32))]


class PdMerged(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("n_samples", C.c_uint32), ("pl", C.POINTER(C.c_uint32)), ("lad", C.POINTER(C.c_uint32)),
                ("dad", C.POINTER(C.c_uint32)), ("first_last", C.POINTER(C.c_uint32)), ("alleles", C.POINTER(C.c_uint32))]


class PdStage1View(C.Structure):
    _fields_ = [("n_windows", C.c_uint32), ("n_samples", C.c_uint32), ("q3", C.POINTER(C.c_int32)),
                ("flags", C.POINTER(C.c_uint8))]


class PdTiming(C.Structure):
    _fields_ = [("stage1_ms", C.c_double), ("stage2_ms", C.c_double), ("n_cells", C.c_uint64),
                ("n_active_pairs", C.c_uint64), ("n_pair_chunks", C.c_uint64), ("n_candidate_cells", C.c_uint64),
                ("max_life", C.c_uint32), ("n_all_gathers", C.c_uint32), ("all_gather_bytes", C.c_uint64),
                ("n_repeated_batches", C.c_uint64), ("exchange_ms", C.c_double),
                ("lik_first_ms", C.c_double), ("lik_first_units", C.c_uint64), ("n_near_integer_shifts", C.c_uint64)]


RESERVE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p)


class PdWalkEntry(C.Structure):
    _fields_ = [("start", C.c_uint32), ("dev", C.c_int32)]


class PdWalk(C.Structure):
    _fields_ = [("first_window_pos", C.c_uint32), ("step_pos", C.c_uint32), ("has_step", C.c_uint32), ("n_windows", C.c_uint32),
                ("window_shift", C.c_uint32), ("n_readgroups", C.c_uint32), ("rg_counts", C.POINTER(C.c_uint32)),
                ("rg_entries", C.POINTER(C.POINTER(PdWalkEntry))), ("end_border", C.POINTER(C.c_uint32)),
                ("id_offset", C.POINTER(C.c_uint32)), ("switches", C.POINTER(C.c_uint8)), ("equal_run", C.POINTER(C.c_uint8)),
                ("rg_sources", C.c_void_p)]


class PdStageTiming(C.Structure):
    _fields_ = [("n_instances", C.c_uint64), ("bytes_uploaded", C.c_uint64), ("pack_seconds", C.c_double), ("uploaded_seconds", C.c_double),
                ("flavour", C.c_uint32), ("host_threads", C.c_uint32), ("counted_seconds", C.c_double), ("total_seconds", C.c_double)]


class PdShard(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("first_sample", C.c_uint32),
                ("n_samples_global", C.c_uint32), ("n_readgroups_global", C.c_uint32),
                ("min_init_del_len_global", C.POINTER(C.c_uint32)), ("reserve", RESERVE_FN),
                ("all_gather", ALL_GATHER_FN), ("user", C.c_void_p)]


def load_library(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    lib = C.CDLL(path)
    lib.pd_ctx_create.argtypes = [C.POINTER(PdConfig), C.POINTER(PdReadgroup), C.c_uint32, C.POINTER(C.c_uint32),
                                  C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    lib.pd_ctx_create.restype = C.c_int
    lib.pd_ctx_destroy.argtypes = [C.c_void_p]
    lib.pd_ctx_destroy.restype = None
    for name in ("pd_segment_submit", "pd_segment_stage"):
        getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(PdSegment), C.POINTER(C.c_uint64)]
        getattr(lib, name).restype = C.c_int
    lib.pd_segment_run.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.pd_segment_run.restype = C.c_int
    lib.pd_segment_unstage.argtypes = [C.c_void_p, C.c_uint64]
    lib.pd_segment_unstage.restype = C.c_int
    lib.pd_segment_collect.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(PdCalls))]
    lib.pd_segment_collect.restype = C.c_int
    lib.pd_calls_free.argtypes = [C.POINTER(PdCalls)]
    lib.pd_calls_free.restype = None
    lib.pd_stage1_last.argtypes = [C.c_void_p, C.POINTER(PdStage1View)]
    lib.pd_stage1_last.restype = C.c_int
    lib.pd_last_timing.argtypes = [C.c_void_p, C.POINTER(PdTiming)]
    lib.pd_last_stage_timing.argtypes = [C.c_void_p, C.POINTER(PdStageTiming)]
    lib.pd_last_stage_timing.restype = C.c_int
    lib.pd_host_copy_gbs.argtypes = [C.c_uint64]
    lib.pd_host_copy_gbs.restype = C.c_double
    lib.pd_ctx_stage1_view.argtypes = [C.c_void_p, C.c_int]
    lib.pd_ctx_stage1_view.restype = C.c_int
    lib.pd_last_timing.restype = C.c_int
    lib.pd_last_error.argtypes = [C.c_void_p]
    lib.pd_last_error.restype = C.c_char_p
    lib.pd_backend_name.restype = C.c_char_p
    lib.pd_abi_version.restype = C.c_int
    lib.pd_shard_samples.argtypes = [C.c_uint32, C.c_uint32]
    lib.pd_shard_samples.restype = C.c_uint32
    lib.pd_ctx_set_shard.argtypes = [C.c_void_p, C.POINTER(PdShard)]
    lib.pd_ctx_set_shard.restype = C.c_int
    lib.pd_ctx_set_shard_rccl.argtypes = [C.c_void_p, C.POINTER(PdShard), C.c_char_p]
    lib.pd_ctx_set_shard_rccl.restype = C.c_int
    lib.pd_log_table_view.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint32)]
    lib.pd_log_table_view.restype = C.c_int
    lib.pd_ctx_keep_results_on_device.argtypes = [C.c_void_p, C.c_int]
    lib.pd_ctx_keep_results_on_device.restype = C.c_int
    lib.pd_calls_merge.argtypes = [C.c_void_p, C.POINTER(PdCalls), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), C.POINTER(C.POINTER(PdMerged))]
    lib.pd_calls_merge.restype = C.c_int
    lib.pd_calls_fetch_columns.argtypes = [C.c_void_p, C.POINTER(PdCalls)]
    lib.pd_calls_fetch_columns.restype = C.c_int
    lib.pd_merged_free.argtypes = [C.POINTER(PdMerged)]
    lib.pd_merged_free.restype = None
    lib.pd_rccl_unique_id.argtypes = [C.c_char_p]
    lib.pd_rccl_unique_id.restype = C.c_int
    lib.pd_walk_submit.argtypes = [C.c_void_p, C.POINTER(PdWalk), C.POINTER(C.c_uint64)]
    lib.pd_walk_submit.restype = C.c_int
    lib.pd_walk_supported.argtypes = [C.c_void_p]
    lib.pd_walk_supported.restype = C.c_int
    lib.pd_walk_active.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_uint32), C.c_uint64]
    lib.pd_walk_active.restype = C.c_int
    lib.pd_device_warmup.argtypes = [C.c_int]
    lib.pd_device_warmup.restype = C.c_int
    return lib


class BackendError(RuntimeError):
    pass


def rccl_unique_id(lib) -> bytes:
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    if lib.pd_rccl_unique_id(buf) != 0:
        raise BackendError(lib.pd_last_error(None).decode())
    return buf.raw


def _let_torch_initialise_first():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so) next to the system one this library
    links (/opt/rocm/lib/libamdhip64.so.7). Both can live in one process, but only if torch's initialises FIRST: the
    other way round torch later reports "No HIP GPUs are available". Processes that never touch torch.cuda are not
    affected; since bench.py, the multi-GPU drivers and some tests do, the binding makes the order right up front
    whenever torch is installed (POPDEL_NO_TORCH_INIT=1 skips this)."""
    if os.environ.get("POPDEL_NO_TORCH_INIT"):
        return
    import importlib.util
    if importlib.util.find_spec("torch") is None:
        return
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()


def offer_host_region(lib, array):
    """pd_host_regions_offer for the memory of `array` (which must stay mapped for the rest of the process)."""
    lib.pd_host_regions_offer.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32]
    lib.pd_host_regions_offer.restype = C.c_int
    begin = (C.c_void_p * 1)(array.ctypes.data)
    size = (C.c_uint64 * 1)(array.nbytes)
    return lib.pd_host_regions_offer(begin, size, 1)


def load_hip():
    """The product backend. Fails loudly when the HIP library has not been built."""
    if not os.path.exists(HIP_LIBRARY):
        raise RuntimeError("libpopdel_hip.so is missing (%s): build it with __graft_entry__.build() or "
                           "`make -C popdel_amd/csrc`; there is no CPU fallback." % HIP_LIBRARY)
    _let_torch_initialise_first()
    lib = load_library(HIP_LIBRARY)
    if lib.pd_backend_name() != b"hip-gfx950":
        raise RuntimeError("unexpected backend behind %s: %r" % (HIP_LIBRARY, lib.pd_backend_name()))
    return lib


@dataclass
class ReadGroup:
    """One conditioned read group (see pd_readgroup)."""
    median: int
    read_length: int
    stddev: float
    min_prob: float
    hist_offset: int
    hist_values: np.ndarray
    lower_quantile_dist: int
    upper_quantile_dist: int
    max_load: int = 100
    min_init_del_len: int = 0


@dataclass
class Segment:
    """Pair instances of one segment (see pd_segment); arrays are read-group major, ordered by ``on``."""
    first_window_pos: int
    n_windows: int
    rg_offsets: np.ndarray
    dev: np.ndarray
    start: np.ndarray
    on: np.ndarray
    off: np.ndarray
    window_shift: int = 30

    def as_struct(self):
        self.rg_offsets = np.ascontiguousarray(self.rg_offsets, dtype=np.uint64)
        self.dev = np.ascontiguousarray(self.dev, dtype=np.int32)
        self.start = np.ascontiguousarray(self.start, dtype=np.uint32)
        self.on = np.ascontiguousarray(self.on, dtype=np.uint32)
        self.off = np.ascontiguousarray(self.off, dtype=np.uint32)
        s = PdSegment()
        s.first_window_pos = self.first_window_pos
        s.n_windows = self.n_windows
        s.window_shift = self.window_shift
        s.n_readgroups = len(self.rg_offsets) - 1
        s.rg_offsets = self.rg_offsets.ctypes.data_as(C.POINTER(C.c_uint64))
        s.pair_dev = self.dev.ctypes.data_as(C.POINTER(C.c_int32))
        s.pair_start = self.start.ctypes.data_as(C.POINTER(C.c_uint32))
        s.pair_on = self.on.ctypes.data_as(C.POINTER(C.c_uint32))
        s.pair_off = self.off.ctypes.data_as(C.POINTER(C.c_uint32))
        return s


class Context:
    """A calling context (pd_ctx) on one device."""

    def __init__(self, lib, readgroups, sample_rg_offsets, iterations=15, min_len=200, min_likelihood_ratio=12.527629,
                 min_sample_fraction=0.1, somatic=False, window_wise_output=False, device=0, stage1_view=False):
        """stage1_view: keep every cell's Q3 for stage1() (pd_ctx_stage1_view: the HIP backend otherwise takes exact Q3 values in
        the flagged windows only; diagnostics and parity tests)."""
        self.lib = lib
        self._keep = []
        n = len(readgroups)
        arr = (PdReadgroup * n)()
        for i, r in enumerate(readgroups):
            hv = np.ascontiguousarray(r.hist_values, dtype=np.float64)
            self._keep.append(hv)
            arr[i].median = r.median
            arr[i].read_length = r.read_length
            arr[i].stddev = r.stddev
            arr[i].min_prob = r.min_prob
            arr[i].hist_offset = r.hist_offset
            arr[i].hist_len = len(hv)
            arr[i].hist_values = hv.ctypes.data_as(C.POINTER(C.c_double))
            arr[i].lower_quantile_dist = r.lower_quantile_dist
            arr[i].upper_quantile_dist = r.upper_quantile_dist
            arr[i].max_load = r.max_load
            arr[i].min_init_del_len = r.min_init_del_len
        cfg = PdConfig(iterations, min_len, min_likelihood_ratio, min_sample_fraction, int(somatic), int(window_wise_output))
        offs = np.ascontiguousarray(sample_rg_offsets, dtype=np.uint32)
        self.n_samples = len(offs) - 1
        self.n_readgroups = n
        handle = C.c_void_p()
        rc = lib.pd_ctx_create(C.byref(cfg), arr, n, offs.ctypes.data_as(C.POINTER(C.c_uint32)), self.n_samples, device,
                               C.byref(handle))
        if rc != 0:
            raise BackendError(lib.pd_last_error(None).decode())
        self.handle = handle
        if stage1_view:
            self._check(lib.pd_ctx_stage1_view(handle, 1))

    def close(self):
        if self.handle:
            self.lib.pd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BackendError(self.lib.pd_last_error(self.handle).decode())

    def set_shard(self, rank, world, first_sample, n_samples_global, min_init_del_len_global, reserve, all_gather):
        """Declare this context one shard of a cohort split by samples (pd_ctx_set_shard). ``reserve(bytes) -> device
        pointer`` and ``all_gather(offset, bytes_per_rank, hip_stream) -> 0`` are Python callables (see
        popdel_amd.distributed.SampleShardExchange)."""
        table = np.ascontiguousarray(min_init_del_len_global, dtype=np.uint32)
        sh = PdShard()
        sh.rank, sh.world, sh.first_sample = rank, world, first_sample
        sh.n_samples_global, sh.n_readgroups_global = n_samples_global, len(table)
        sh.min_init_del_len_global = table.ctypes.data_as(C.POINTER(C.c_uint32))
        self._shard_fns = (RESERVE_FN(lambda user, nbytes: reserve(nbytes)),
                           ALL_GATHER_FN(lambda user, off, per, stream: all_gather(off, per, stream)))
        sh.reserve, sh.all_gather = self._shard_fns
        sh.user = None
        self._check(self.lib.pd_ctx_set_shard(self.handle, C.byref(sh)))

    def set_shard_lanes(self, rank, world, first_sample, n_samples_global, min_init_del_len_global, channels):
        """pd_ctx_set_shard_lanes: the context keeps its lanes in a cohort over several ranks. ``channels``: a list of
        (reserve, all_gather) pairs of Python callables -- the first for the context itself, one more per lane (each with its
        own buffer and group: popdel_amd.distributed.SampleShardExchange)."""
        table = np.ascontiguousarray(min_init_del_len_global, dtype=np.uint32)
        self._shard_fns = []
        structs = (PdShard * len(channels))()
        for k, (reserve, all_gather) in enumerate(channels):
            fns = (RESERVE_FN(lambda user, nbytes, f=reserve: f(nbytes)),
                   ALL_GATHER_FN(lambda user, off, per, stream, f=all_gather: f(off, per, stream)))
            self._shard_fns.append(fns)
            sh = structs[k]
            sh.rank, sh.world, sh.first_sample = rank, world, first_sample
            sh.n_samples_global, sh.n_readgroups_global = n_samples_global, len(table)
            sh.min_init_del_len_global = table.ctypes.data_as(C.POINTER(C.c_uint32))
            sh.reserve, sh.all_gather = fns
            sh.user = None
        self.lib.pd_ctx_set_shard_lanes.restype = C.c_int
        self._check(self.lib.pd_ctx_set_shard_lanes(self.handle, C.byref(structs[0]), C.cast(C.byref(structs, C.sizeof(PdShard)), C.POINTER(PdShard)),
                                                    C.c_uint32(len(channels) - 1)))

    def set_shard_rccl(self, rank, world, first_sample, n_samples_global, min_init_del_len_global, rccl_id: bytes):
        """The same with the exchange done by the library itself over RCCL (pd_ctx_set_shard_rccl); collective: every
        context of the cohort calls it with the 128 bytes one of them got from ``rccl_unique_id``."""
        table = np.ascontiguousarray(min_init_del_len_global, dtype=np.uint32)
        sh = PdShard()
        sh.rank, sh.world, sh.first_sample = rank, world, first_sample
        sh.n_samples_global, sh.n_readgroups_global = n_samples_global, len(table)
        sh.min_init_del_len_global = table.ctypes.data_as(C.POINTER(C.c_uint32))
        assert len(rccl_id) == RCCL_ID_BYTES
        self._check(self.lib.pd_ctx_set_shard_rccl(self.handle, C.byref(sh), rccl_id))

    def submit(self, seg: Segment):
        t = C.c_uint64()
        st = seg.as_struct()
        self._check(self.lib.pd_segment_submit(self.handle, C.byref(st), C.byref(t)))
        return t.value

    def submit_walk(self, first_window_pos, n_windows, rg_starts, rg_devs, end_border, switches, equal_run, step_pos=None,
                    id_offset=None, window_shift=30, table_arena=None, in_arena=None):
        """pd_walk_submit: the segment as the loader's tables hold it. rg_starts / rg_devs: per read group the start
        positions and deviations of its new start-slot entries (arrays; ignored where switches[rg] == 0). table_arena: a
        (N, 2) uint32 array the tables are laid out in one behind the other (a stand-in for the host's table pool, see
        offer_host_region); in_arena: which read groups go there (default: all)."""
        n_rg = len(rg_starts)
        tables = [np.ascontiguousarray(np.stack([np.asarray(s, np.uint32), np.asarray(d, np.int32).view(np.uint32)], axis=1))
                  if len(s) else np.zeros((0, 2), np.uint32) for s, d in zip(rg_starts, rg_devs)]
        if table_arena is not None:
            at = 1                          # (an odd entry: runs begin on any 8-byte boundary)
            for r in range(n_rg):
                if len(tables[r]) and (in_arena is None or in_arena[r]):
                    view = table_arena[at:at + len(tables[r])]
                    view[:] = tables[r]
                    tables[r] = view
                    at += len(view) + (r % 2)
        counts = np.array([len(t) for t in tables], np.uint32)
        ptrs = (C.POINTER(PdWalkEntry) * n_rg)(*[t.ctypes.data_as(C.POINTER(PdWalkEntry)) for t in tables])
        border = np.ascontiguousarray(end_border, np.uint32)
        ids = np.ascontiguousarray(id_offset if id_offset is not None else np.zeros(n_rg), np.uint32)
        sw = np.ascontiguousarray(switches, np.uint8)
        run = np.ascontiguousarray(equal_run, np.uint8)
        wk = PdWalk(first_window_pos, step_pos or 0, 0 if step_pos is None else 1, n_windows, window_shift, n_rg,
                    counts.ctypes.data_as(C.POINTER(C.c_uint32)), ptrs, border.ctypes.data_as(C.POINTER(C.c_uint32)),
                    ids.ctypes.data_as(C.POINTER(C.c_uint32)), sw.ctypes.data_as(C.POINTER(C.c_uint8)),
                    run.ctypes.data_as(C.POINTER(C.c_uint8)), None)
        t = C.c_uint64()
        self._check(self.lib.pd_walk_submit(self.handle, C.byref(wk), C.byref(t)))
        return t.value

    def walk_entry_counts(self):
        """(entries the device fetched from offered host regions, entries staged by the host threads, bytes page-locked)"""
        g, st, lk = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.pd_walk_entry_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self._check(self.lib.pd_walk_entry_counts(self.handle, C.byref(g), C.byref(st), C.byref(lk)))
        return g.value, st.value, lk.value

    def walk_active(self):
        """(read group, start, deviation, id) arrays of the pairs still active after the last submitted walk"""
        n = C.c_uint64()
        self._check(self.lib.pd_walk_active(self.handle, C.byref(n), None, None, None, None, 0))
        k = n.value
        rg, st, dv, ids = np.zeros(k, np.uint32), np.zeros(k, np.uint32), np.zeros(k, np.int32), np.zeros(k, np.uint32)
        if k:
            self._check(self.lib.pd_walk_active(self.handle, C.byref(n), rg.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                st.ctypes.data_as(C.POINTER(C.c_uint32)), dv.ctypes.data_as(C.POINTER(C.c_int32)),
                                                ids.ctypes.data_as(C.POINTER(C.c_uint32)), k))
        return rg, st, dv, ids

    def stage(self, seg: Segment):
        h = C.c_uint64()
        st = seg.as_struct()
        self._check(self.lib.pd_segment_stage(self.handle, C.byref(st), C.byref(h)))
        return h.value

    def run(self, handle):
        t = C.c_uint64()
        self._check(self.lib.pd_segment_run(self.handle, handle, C.byref(t)))
        return t.value

    def unstage(self, handle):
        self._check(self.lib.pd_segment_unstage(self.handle, handle))

    def collect(self, ticket, copy=True):
        """Returns a dict of numpy arrays (copies). With copy=False only the call count is returned and the result
        block is released right away (throughput runs: the results have reached host memory, nobody reads them)."""
        p = C.POINTER(PdCalls)()
        self._check(self.lib.pd_segment_collect(self.handle, ticket, C.byref(p)))
        c = p.contents
        n, S = c.n_calls, c.n_samples
        if not copy:
            self.lib.pd_calls_free(p)
            return dict(n_calls=n, n_samples=S)

        def arr(ptr, shape, dtype):
            count = int(np.prod(shape)) if len(shape) else 1
            if count == 0:
                return np.zeros(shape, dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype).reshape(shape).copy()

        out = dict(
            n_calls=n, n_samples=S,
            initial_length=arr(c.initial_length, (n,), np.uint32), iterations=arr(c.iterations, (n,), np.uint32),
            deletion_length=arr(c.deletion_length, (n,), np.uint32),
            likelihood_ratio=arr(c.likelihood_ratio, (n,), np.float64), frequency=arr(c.frequency, (n,), np.float64),
            window_position=arr(c.window_position, (n,), np.uint32), position=arr(c.position, (n,), np.uint32),
            end_position=arr(c.end_position, (n,), np.uint32), filter=arr(c.filter, (n,), np.uint8),
            pl=arr(c.pl, (n, S, 3), np.uint32), lad=arr(c.lad, (n, S, 3), np.uint32), dad=arr(c.dad, (n, S, 5), np.uint32),
            first_last=arr(c.first_last, (n, S, 2), np.uint32),
            near_integer_shifts=arr(c.near_integer_shifts, (n,), np.uint32))
        self.lib.pd_calls_free(p)
        return out

    # ---- merging on the device (pd_ctx_keep_results_on_device / pd_calls_merge) -------------------------------------------
    def keep_results_on_device(self, on=True):
        self._check(self.lib.pd_ctx_keep_results_on_device(self.handle, int(on)))

    def collect_raw(self, ticket):
        """The pd_calls pointer itself (release with ``free_raw``): what pd_calls_merge / pd_calls_fetch_columns take."""
        p = C.POINTER(PdCalls)()
        self._check(self.lib.pd_segment_collect(self.handle, ticket, C.byref(p)))
        return p

    def free_raw(self, p):
        self.lib.pd_calls_free(p)

    def fetch_columns(self, p):
        self._check(self.lib.pd_calls_fetch_columns(self.handle, p))

    def merge(self, p, groups, heads):
        """groups: list of lists of call indices; heads: one call index per group. Returns a dict of numpy arrays, or None
        if the device declined (return code 2)."""
        offsets = np.zeros(len(groups) + 1, dtype=np.uint32)
        np.cumsum([len(g) for g in groups], out=offsets[1:])
        members = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.uint32) for g in groups]) if groups else np.zeros(0, np.uint32))
        heads = np.ascontiguousarray(heads, dtype=np.uint32)
        out = C.POINTER(PdMerged)()
        u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        rc = self.lib.pd_calls_merge(self.handle, p, len(groups), u32(offsets), u32(members), u32(heads), C.byref(out))
        if rc == 2:
            return None
        self._check(rc)
        m = out.contents
        G, S = m.n_groups, m.n_samples

        def arr(ptr, k):
            if G * S == 0:
                return np.zeros((G, S, k), np.uint32)
            return np.ctypeslib.as_array(ptr, shape=(G * S * k,)).reshape(G, S, k).copy()

        res = dict(pl=arr(m.pl, 3), lad=arr(m.lad, 3), dad=arr(m.dad, 5), first_last=arr(m.first_last, 2),
                   alleles=np.ctypeslib.as_array(m.alleles, shape=(G,)).copy() if G else np.zeros(0, np.uint32))
        self.lib.pd_merged_free(out)
        return res

    def stage1(self):
        v = PdStage1View()
        self._check(self.lib.pd_stage1_last(self.handle, C.byref(v)))
        n = v.n_windows * v.n_samples
        if n == 0:
            return np.zeros((0, 0), np.int32), np.zeros((0, 0), np.uint8)
        q3 = np.ctypeslib.as_array(v.q3, shape=(n,)).reshape(v.n_windows, v.n_samples).copy()
        fl = np.ctypeslib.as_array(v.flags, shape=(n,)).reshape(v.n_windows, v.n_samples).copy()
        return q3, fl

    def log_tables(self, rg, hist_len):
        """(values, ln, log10) of read group `rg` as the backend holds them (pd_log_table_view)."""
        v, l, l10 = (np.zeros(hist_len, np.float64) for _ in range(3))
        n = C.c_uint32()
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self.lib.pd_log_table_view(self.handle, rg, p(v), p(l), p(l10), C.byref(n)))
        return v[:n.value], l[:n.value], l10[:n.value]

    def stage_timing(self):
        """Host and link cost of the context's last stage() / submit() (pd_last_stage_timing)."""
        t = PdStageTiming()
        self._check(self.lib.pd_last_stage_timing(self.handle, C.byref(t)))
        return {k: getattr(t, k) for k, _ in PdStageTiming._fields_}

    def timing(self):
        t = PdTiming()
        self._check(self.lib.pd_last_timing(self.handle, C.byref(t)))
        return {k: getattr(t, k) for k, _ in PdTiming._fields_}
