"""ctypes binding of the C ABI in include/centrifuge_amd.h (libcentrifuge_amd.so).

Plumbing for the tests and bench.py; the product is the shared library and the
C++ front end built from centrifuge_amd/csrc.  There is no CPU fallback here:
if the library (the HIP extension) is missing, importing `lib()` raises.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CF_AMD_LIB") or os.path.join(HERE, "libcentrifuge_amd.so")   # the env knob: A/B builds in tools/ experiments

ROW_DTYPE = np.dtype([("tax_id", "<u8"), ("unique_id", "<u4"), ("score", "<u4"), ("hit_len", "<u4"),
                      ("taxon_idx", "<u4")])
HIT_DTYPE = np.dtype([("top", "<u8"), ("bot", "<u8"), ("bwoff", "<u4"), ("len", "<u4")])
RANK_SLOTS = {"strain": 0, "species": 1, "genus": 2, "family": 3, "order": 4, "class": 5, "phylum": 6}
MERGED = 0xffffffff


class Params(C.Structure):
    _fields_ = [("khits", C.c_int32), ("min_hitlen", C.c_int32), ("rank_slot", C.c_int32),
                ("tree_traverse", C.c_int32),
                ("host_taxids", C.POINTER(C.c_uint64)), ("n_host", C.c_int32),
                ("exclude_taxids", C.POINTER(C.c_uint64)), ("n_exclude", C.c_int32)]


class OpCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_ftab", "n_pair", "n_pair2", "n_single", "n_walk", "n_rows", "n_ftab_wide", "n_verify", "n_text_loads", "n_pos_hits")]

    def sides(self):
        return self.n_pair + self.n_pair2 + self.n_single + self.n_walk

    def algorithmic_bytes(self, sa_bytes, n_reads, read_len, step_bytes=128, walk_step_bytes=128):
        """Bytes this batch's kernels must touch, each request at its own granule: an LF step reads one 128-byte side
         or, with the occurrence planes, one 16-byte entry; a wide-ftab entry, an SA / inverse-SA sample 8,
        a 10-mer ftab pair 16, a text window 32, a resolved row its SA-sample entry, plus the packed read in and the row out."""
        per_read = (read_len + 3) // 4 + (read_len + 7) // 8 + 32
        return step_bytes * (self.n_pair + self.n_pair2 + self.n_single) + walk_step_bytes * self.n_walk + \
            16 * self.n_ftab + 8 * (self.n_ftab_wide + 2 * self.n_verify - self.n_pos_hits) + 32 * self.n_text_loads + sa_bytes * self.n_rows + per_read * n_reads


class PackedReads(C.Structure):
    """cf_packed_reads of include/centrifuge_amd.h"""
    _fields_ = [("bases", C.c_void_p), ("nmask", C.c_void_p), ("len", C.c_void_p), ("seeds", C.c_void_p),
                ("n_reads", C.c_uint64), ("n_words", C.c_uint64), ("n_bases", C.c_uint64),
                ("max_len", C.c_uint32), ("paired", C.c_int32),
                ("nword_idx", C.c_void_p), ("nword_mask", C.c_void_p), ("n_nwords", C.c_uint64)]


def sparse_nmask(nmask):
    """(indices, masks) of the N-mask words that are not zero: the sparse form of cf_packed_reads"""
    idx = np.flatnonzero(nmask).astype(np.uint64)
    return idx, np.ascontiguousarray(nmask[idx], dtype=np.uint32)


class Results(C.Structure):
    """cf_results of include/centrifuge_amd.h"""
    _fields_ = [("rows", C.c_void_p), ("n_rows", C.c_void_p), ("score2", C.c_void_p), ("max_score", C.c_void_p),
                ("n_queries", C.c_uint64), ("total_rows", C.c_uint64), ("planned_sa_rows", C.c_uint64),
                ("row_passes", C.c_uint32), ("slow_post", C.c_uint32), ("slow_score", C.c_uint32)]


class DenseReads(C.Structure):
    """cf_dense_reads of include/centrifuge_amd.h"""
    _fields_ = [("bases4", C.c_void_p), ("seeds", C.c_void_p), ("n_reads", C.c_uint64), ("read_len", C.c_uint32), ("paired", C.c_int32),
                ("nword_idx", C.c_void_p), ("nword_mask", C.c_void_p), ("n_nwords", C.c_uint64)]


class ResultsNarrow(C.Structure):
    """cf_results_narrow of include/centrifuge_amd.h"""
    _fields_ = [("rows", C.c_void_p), ("qinfo", C.c_void_p), ("score2", C.c_void_p),
                ("n_queries", C.c_uint64), ("total_rows", C.c_uint64), ("planned_sa_rows", C.c_uint64),
                ("row_passes", C.c_uint32), ("slow_post", C.c_uint32), ("slow_score", C.c_uint32)]


class TextReads(C.Structure):
    """cf_text_reads of include/centrifuge_amd.h"""
    _fields_ = [("text", C.c_void_p), ("n_bytes", C.c_uint64), ("format", C.c_int32), ("global_seed", C.c_uint32), ("max_reads", C.c_uint64),
                ("text2", C.c_void_p), ("n_bytes2", C.c_uint64)]


class TextInfo(C.Structure):
    """cf_text_info of include/centrifuge_amd.h"""
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("max_len", C.c_uint32), ("irregular", C.c_uint32)]


class ResultsText(C.Structure):
    """cf_results_text of include/centrifuge_amd.h"""
    _fields_ = [("text", C.c_void_p), ("n_bytes", C.c_uint64), ("tuples", C.c_void_p), ("n_tuple_words", C.c_uint64),
                ("n_queries", C.c_uint64), ("total_rows", C.c_uint64), ("planned_sa_rows", C.c_uint64),
                ("row_passes", C.c_uint32), ("slow_post", C.c_uint32), ("slow_score", C.c_uint32)]


TEXT_FASTA, TEXT_FASTQ = 0, 1
ROW16_DTYPE = np.dtype([("unique_id", "<u4"), ("taxon_idx", "<u4"), ("score", "<u4"), ("hit_len", "<u4")])
RESULTS_ROWS, RESULTS_NARROW = 0, 1


def dense_pack(codes):
    """[n_reads, L] base codes 0..4 -> (bases4: n_reads * ceil(L/4) bytes, four bases per byte, an N as code 0; word indices and
    mask words of the sparse N mask in the device's word numbering: read r owns words [r * ceil(L/32), ...))"""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, L = codes.shape
    bpr, W = (L + 3) // 4, (L + 31) // 32
    pad = np.zeros((n, bpr * 4), dtype=np.uint8)
    pad[:, :L] = np.where(codes > 3, 0, codes)
    q = pad.reshape(n, bpr, 4)
    b4 = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8).reshape(-1)
    r, i = np.nonzero(codes > 3)
    word = r.astype(np.uint64) * np.uint64(W) + (i // 32).astype(np.uint64)
    idx, inv = np.unique(word, return_inverse=True)
    mask = np.zeros(len(idx), dtype=np.uint32)
    np.bitwise_or.at(mask, inv, (np.uint32(1) << (i % 32).astype(np.uint32)))
    return b4, idx.astype(np.uint64), mask


class BuildInput(C.Structure):
    """cf_build_input of include/centrifuge_amd_build.h"""
    _fields_ = [("fasta_paths", C.POINTER(C.c_char_p)), ("n_fasta", C.c_int32),
                ("codes", C.c_void_p), ("seq_off", C.c_void_p), ("seq_names", C.POINTER(C.c_char_p)), ("n_seq", C.c_uint64),
                ("conversion_table", C.c_char_p), ("taxonomy_tree", C.c_char_p), ("name_table", C.c_char_p),
                ("size_table", C.c_char_p), ("off_rate", C.c_int32), ("ftab_chars", C.c_int32),
                ("chunk_suffixes", C.c_uint64), ("verbose", C.c_int32)]


def make_params(k=5, min_hitlen=22, rank="strain", traverse=True, host=(), exclude=()):
    p = Params()
    p.khits, p.min_hitlen, p.rank_slot, p.tree_traverse = k, min_hitlen, RANK_SLOTS[rank], int(traverse)
    p._host = (C.c_uint64 * max(1, len(host)))(*host)
    p._excl = (C.c_uint64 * max(1, len(exclude)))(*exclude)
    p.host_taxids, p.n_host = p._host, len(host)
    p.exclude_taxids, p.n_exclude = p._excl, len(exclude)
    return p


EXPORTS = [
    "cf_strerror", "cf_last_error", "cf_index_open", "cf_index_open_ex", "cf_index_describe", "cf_debug_plan_tables", "cf_index_open_host", "cf_index_close", "cf_index_text_len",
    "cf_index_num_refs", "cf_index_num_taxa", "cf_index_device_bytes", "cf_index_compressed", "cf_index_sa_width",
    "cf_index_uid", "cf_index_ref_taxid", "cf_index_taxon_id", "cf_format_seqid", "cf_tax_rank",
    "cf_tax_rank_string", "cf_tax_name", "cf_tax_size", "cf_params_default", "cf_classifier_create",
    "cf_classifier_destroy", "cf_batch_create", "cf_batch_destroy", "cf_batch_num_queries", "cf_gen_rand_seed",
    "cf_classify", "cf_batch_results", "cf_batch_timings", "cf_batch_opcounts", "cf_counts_reset", "cf_counts_get",
    "cf_counts_device", "cf_counts_allreduce", "cf_debug_search", "cf_debug_resolve", "cf_debug_rank", "cf_debug_rank1",
    "cf_debug_random_read_gbps", "cf_index_restore", "cf_batch_num_rows", "cf_batch_results_compact", "cf_batch_plan", "cf_batch_plan_ms",
    "cf_batch_max_scores", "cf_report_create", "cf_report_destroy", "cf_report_add", "cf_report_add_narrow", "cf_report_add_counts", "cf_report_reset_counts", "cf_report_write", "cf_report_serialize", "cf_report_merge",
    "cf_index_text_verify_rate", "cf_index_text_verify_build_ms", "cf_index_wide_ftab_chars", "cf_index_occ_planes", "cf_index_occ_planes_build_ms", "cf_index_resolve_rate", "cf_index_resolve_build_ms", "cf_index_walk_bound", "cf_index_resolve_by_position", "cf_slot_estimate_bytes", "cf_batch_reclassify_async", "cf_comm_init_all", "cf_comm_destroy", "cf_counts_allreduce_group", "cf_stream_create", "cf_stream_destroy", "cf_device_count", "cf_device_numa_node", "cf_thread_bind_near_device",
    "cf_report_adopt_counts", "cf_debug_scan", "cf_host_alloc", "cf_host_free", "cf_batch_alloc", "cf_batch_upload_packed_async", "cf_classify_async", "cf_batch_download_async",
    "cf_batch_submit", "cf_batch_wait", "cf_batch_upload", "cf_batch_set_limits",
    "cf_batch_upload_dense_async", "cf_batch_set_result_format", "cf_batch_wait_narrow", "cf_narrow_max_score", "cf_results_narrow_expand",
    "cf_build_input_default", "cf_build_index", "cf_build_timings", "cf_build_last_error", "cf_build_taxonomy", "cf_build_describe",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libcentrifuge_amd.so (the HIP extension) is not built: run __graft_entry__.build(); "
                           "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, cp = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p
    sig = {
        "cf_strerror": (cp, [i32]), "cf_last_error": (cp, []),
        "cf_index_open": (i32, [cp, i32, C.POINTER(vp)]), "cf_index_open_host": (i32, [cp, C.POINTER(vp)]),
        "cf_index_open_ex": (i32, [cp, i32, vp, C.POINTER(vp)]), "cf_index_describe": (i32, [vp, vp]),
        "cf_debug_plan_tables": (i32, [u64, i32, i32, i32, u64, vp, vp, vp, vp]),
        "cf_index_close": (None, [vp]),
        "cf_index_text_len": (u64, [vp]), "cf_index_num_refs": (u64, [vp]), "cf_index_num_taxa": (u64, [vp]),
        "cf_index_device_bytes": (u64, [vp]), "cf_index_compressed": (i32, [vp]), "cf_index_sa_width": (i32, [vp]),
        "cf_index_uid": (cp, [vp, u64]), "cf_index_ref_taxid": (u64, [vp, u64]), "cf_index_taxon_id": (u64, [vp, u64]),
        "cf_format_seqid": (cp, [vp, u32, u64]), "cf_tax_rank": (i32, [vp, u64]), "cf_tax_rank_string": (cp, [i32]),
        "cf_tax_name": (cp, [vp, u64]), "cf_tax_size": (u64, [vp, u64]),
        "cf_params_default": (i32, [C.POINTER(Params)]),
        "cf_classifier_create": (i32, [vp, C.POINTER(Params), C.POINTER(vp)]), "cf_classifier_destroy": (None, [vp]),
        "cf_batch_create": (i32, [vp, vp, vp, vp, u64, i32, C.POINTER(vp)]), "cf_batch_destroy": (None, [vp]),
        "cf_batch_num_queries": (u64, [vp]),
        "cf_gen_rand_seed": (u32, [vp, vp, u64, cp, u64, u32]),
        "cf_classify": (i32, [vp, vp, vp]),
        "cf_batch_results": (i32, [vp, vp, vp, vp]),
        "cf_batch_plan": (i32, [vp, vp]), "cf_batch_plan_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "cf_batch_num_rows": (i32, [vp, C.POINTER(u64)]), "cf_batch_results_compact": (i32, [vp, vp, u64, vp, vp]),
        "cf_batch_timings": (i32, [vp, C.POINTER(C.c_float * 5)]),
        "cf_batch_opcounts": (i32, [vp, C.POINTER(OpCounts)]),
        "cf_counts_reset": (i32, [vp]), "cf_counts_get": (i32, [vp, vp, vp]), "cf_counts_device": (vp, [vp]), "cf_counts_allreduce": (i32, [vp, vp, vp]),
        "cf_debug_search": (i32, [vp, vp, u64, vp, vp, u32, vp]),
        "cf_debug_resolve": (i32, [vp, vp, u64, vp]),
        "cf_debug_rank": (i32, [vp, vp, vp, u64, vp]), "cf_debug_rank1": (i32, [vp, vp, vp, u64, vp]),
        "cf_debug_random_read_gbps": (i32, [vp, u64, i32, C.POINTER(C.c_double)]),
        "cf_index_restore": (i32, [vp, vp, u64]),
        "cf_batch_max_scores": (i32, [vp, vp]),
        "cf_report_create": (i32, [vp, C.POINTER(vp)]), "cf_report_destroy": (None, [vp]),
        "cf_report_add": (i32, [vp, vp, vp, vp, u64, u32]), "cf_report_add_narrow": (i32, [vp, vp, vp, vp, u32, i32, u64]), "cf_report_add_counts": (i32, [vp, vp, vp, vp, u64]), "cf_report_reset_counts": (i32, [vp]),
        "cf_report_write": (i32, [vp, cp, i32, C.POINTER(u64), C.POINTER(C.c_double)]),
        "cf_report_serialize": (i32, [vp, vp, u64, C.POINTER(u64)]), "cf_report_merge": (i32, [vp, vp, u64]),
        "cf_debug_scan": (i32, [i32, i32, vp, u64, vp, vp]),
        "cf_index_text_verify_rate": (i32, [vp]), "cf_index_text_verify_build_ms": (C.c_double, [vp]), "cf_index_wide_ftab_chars": (i32, [vp]), "cf_index_occ_planes": (i32, [vp]), "cf_index_occ_planes_build_ms": (C.c_double, [vp]), "cf_index_resolve_rate": (i32, [vp]), "cf_index_resolve_build_ms": (C.c_double, [vp]), "cf_index_walk_bound": (C.c_uint32, [vp]), "cf_index_resolve_by_position": (i32, [vp]),
        "cf_batch_reclassify_async": (i32, [vp, vp, vp]),
        "cf_slot_estimate_bytes": (i32, [u64, u64, i32, i32, i32, C.POINTER(u64)]),
        "cf_comm_init_all": (i32, [i32, vp, vp]), "cf_comm_destroy": (None, [vp]), "cf_counts_allreduce_group": (i32, [vp, vp, i32]),
        "cf_stream_create": (i32, [i32, C.POINTER(vp)]), "cf_stream_destroy": (None, [vp]), "cf_device_count": (i32, []), "cf_device_numa_node": (i32, [i32]), "cf_thread_bind_near_device": (i32, [i32, C.POINTER(i32)]),
        "cf_report_adopt_counts": (i32, [vp, vp, vp, u64]),
        "cf_host_alloc": (i32, [C.POINTER(vp), C.c_size_t]), "cf_host_free": (None, [vp]),
        "cf_batch_alloc": (i32, [vp, u64, u64, C.POINTER(vp)]),
        "cf_batch_upload_packed_async": (i32, [vp, C.POINTER(PackedReads), vp]),
        "cf_classify_async": (i32, [vp, vp, vp]), "cf_batch_download_async": (i32, [vp, vp]),
        "cf_batch_submit": (i32, [vp, C.POINTER(PackedReads), vp]),
        "cf_batch_wait": (i32, [vp, C.POINTER(Results)]),
        "cf_batch_upload_dense_async": (i32, [vp, C.POINTER(DenseReads), vp]),
        "cf_batch_set_result_format": (i32, [vp, i32]),
        "cf_batch_wait_narrow": (i32, [vp, C.POINTER(ResultsNarrow)]),
        "cf_narrow_max_score": (C.c_uint32, [C.c_uint8, C.c_uint32, C.c_uint32, i32]),
        "cf_results_narrow_expand": (i32, [vp, C.POINTER(ResultsNarrow), vp, C.c_uint32, i32, vp, vp, vp]),
        "cf_batch_upload": (i32, [vp, vp, vp, vp, u64, i32, vp]),
        "cf_batch_upload_text": (i32, [vp, C.POINTER(TextReads), vp, C.POINTER(TextInfo)]),
        "cf_batch_wait_text": (i32, [vp, C.POINTER(ResultsText)]),
        "cf_counts_get_single": (i32, [vp, vp]),
        "cf_report_add_tuples": (i32, [vp, vp, u64]), "cf_report_adopt_device_tally": (i32, [vp, vp, vp, vp, u64]),
        "cf_batch_set_limits": (i32, [vp, u64, u64]),
        "cf_build_input_default": (i32, [C.POINTER(BuildInput)]),
        "cf_build_index": (i32, [C.POINTER(BuildInput), cp, i32]),
        "cf_build_timings": (i32, [C.POINTER(C.c_double * 4)]),
        "cf_build_last_error": (cp, []),
        "cf_build_taxonomy": (i32, [C.POINTER(BuildInput), cp, C.c_char_p, u64]),
        "cf_build_describe": (i32, [C.POINTER(BuildInput), cp, C.c_char_p, u64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


class CfError(RuntimeError):
    pass


def _check(st):
    if st != 0:
        L = lib()
        raise CfError("%s: %s" % (L.cf_strerror(st).decode(), L.cf_last_error().decode()))


class IndexOptions(C.Structure):
    """cf_index_options of include/centrifuge_amd.h"""
    _fields_ = [("hbm_budget_bytes", C.c_uint64), ("wide_ftab_chars", C.c_int32), ("text_verify_rate", C.c_int32),
                ("occ_planes", C.c_int32), ("resolve_rate", C.c_int32), ("pair_planes", C.c_int32), ("sides", C.c_int32),
                ("small_range_rows", C.c_int32), ("reserved_", C.c_int32), ("expected_reads", C.c_uint64)]


class IndexConfig(C.Structure):
    """cf_index_config of include/centrifuge_amd.h"""
    _fields_ = [("text_len", C.c_uint64), ("budget_bytes", C.c_uint64), ("file_section_bytes", C.c_uint64),
                ("wide_ftab_bytes", C.c_uint64), ("wide_ftab_chars", C.c_int32),
                ("text_bytes", C.c_uint64), ("text_verify_rate", C.c_int32),
                ("planes_bytes", C.c_uint64), ("occ_planes", C.c_int32),
                ("pair_planes_bytes", C.c_uint64), ("pair_planes", C.c_int32),
                ("resolve_bytes", C.c_uint64), ("resolve_rate", C.c_int32), ("sides_dropped", C.c_int32),
                ("total_bytes", C.c_uint64), ("build_ms", C.c_double), ("est_requests_per_100bp_read", C.c_double),
                ("file_bytes_dropped", C.c_uint64), ("small_range_rows", C.c_int32), ("plan_realised", C.c_int32),
                ("repeat_fraction", C.c_double)]


def slot_bytes(max_reads, max_words, k=5, ftab_chars=10, occ_planes=True):
    """device bytes of a Slot(clf, max_reads, max_words) (cf_slot_estimate_bytes: no device needed)"""
    out = C.c_uint64()
    _check(lib().cf_slot_estimate_bytes(int(max_reads), int(max_words), int(k), int(ftab_chars), 1 if occ_planes else 0, C.byref(out)))
    return out.value


def plan_tables(n, room, ftab_chars=10, off_rate=4, sa_width=2, **opts):
    """the table planner of cf_index_open on its own (no device) -> dict(K, text_rate, planes, resolve_rate, pair, cost, bytes)"""
    o = IndexOptions(**{k: int(v) for k, v in opts.items()})
    out = (C.c_int32 * 6)()
    cost, nb = C.c_double(), C.c_uint64()
    _check(lib().cf_debug_plan_tables(int(n), ftab_chars, off_rate, sa_width, int(room), C.byref(o), out, C.byref(cost), C.byref(nb)))
    return dict(K=out[0], text_rate=out[1], planes=out[2], resolve_rate=out[3], pair=out[4], drop_sides=out[5], cost=cost.value, bytes=nb.value)


class Index:
    def __init__(self, basename, device=0, host_only=False, hbm_budget=0, **opts):
        """hbm_budget (bytes, 0 = what the device has free) and wide_ftab_chars / text_verify_rate / occ_planes / resolve_rate
        (cf_index_options: 0 = automatic, -1 = off) go through cf_index_open_ex"""
        self.L = lib()
        h = C.c_void_p()
        # the test suite's switch (tests/conftest.py documents it): every index that does not say otherwise is opened with small
        # ranges finished against the text, so that the whole GPU suite can run over that search path as well
        # (behind the same gate as the library's own knobs, cf_knobs.hpp: a stray variable in a user's environment changes nothing)
        if not host_only and "small_range_rows" not in opts and os.environ.get("CF_DEBUG_KNOBS", "0") not in ("", "0") and os.environ.get("CF_TEST_SMALL_RANGE_ROWS"):
            opts["small_range_rows"] = int(os.environ["CF_TEST_SMALL_RANGE_ROWS"])
        if host_only:
            _check(self.L.cf_index_open_host(basename.encode(), C.byref(h)))
        elif hbm_budget or opts:
            o = IndexOptions(hbm_budget_bytes=int(hbm_budget), **{k: int(v) for k, v in opts.items()})
            _check(self.L.cf_index_open_ex(basename.encode(), device, C.byref(o), C.byref(h)))
        else:
            _check(self.L.cf_index_open(basename.encode(), device, C.byref(h)))
        self.h = h

    def describe(self):
        """what the index occupies on the device and which derived tables were made (cf_index_describe) -> dict"""
        c = IndexConfig()
        _check(self.L.cf_index_describe(self.h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in IndexConfig._fields_}

    def close(self):
        if self.h:
            self.L.cf_index_close(self.h)
            self.h = None

    text_len = property(lambda s: s.L.cf_index_text_len(s.h))
    num_refs = property(lambda s: s.L.cf_index_num_refs(s.h))
    num_taxa = property(lambda s: s.L.cf_index_num_taxa(s.h))
    device_bytes = property(lambda s: s.L.cf_index_device_bytes(s.h))
    sa_width = property(lambda s: s.L.cf_index_sa_width(s.h))

    def seqid(self, unique_id, tax_id):
        return self.L.cf_format_seqid(self.h, int(unique_id), int(tax_id)).decode("latin1")

    def taxon_ids(self):
        return np.array([self.L.cf_index_taxon_id(self.h, i) for i in range(self.num_taxa)], dtype=np.uint64)

    def restore(self):
        """The joined text out of the BWT (cf_index_restore): 2-bit packed, text_len/4 + 1 bytes."""
        out = np.zeros(self.text_len // 4 + 1, dtype=np.uint8)
        _check(self.L.cf_index_restore(self.h, out.ctypes.data, out.size))
        return out

    def random_read_gbps(self, n_loads=1 << 26, steps=64):
        g = C.c_double()
        _check(self.L.cf_debug_random_read_gbps(self.h, n_loads, steps, C.byref(g)))
        return g.value

    def debug_rank(self, chars, rows, single_lane=False):
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(len(rows), dtype=np.uint64)
        f = self.L.cf_debug_rank1 if single_lane else self.L.cf_debug_rank
        _check(f(self.h, chars.ctypes.data, rows.ctypes.data, len(rows), out.ctypes.data))
        return out

    def debug_resolve(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(len(rows), dtype=np.uint32)
        _check(self.L.cf_debug_resolve(self.h, rows.ctypes.data, len(rows), out.ctypes.data))
        return out


class Classifier:
    def __init__(self, index, **kw):
        self.index, self.L = index, index.L
        self.params = make_params(**kw)
        h = C.c_void_p()
        _check(self.L.cf_classifier_create(index.h, C.byref(self.params), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.L.cf_classifier_destroy(self.h)
            self.h = None

    def batch(self, seq, off, seeds, paired=False):
        return Batch(self, seq, off, seeds, paired)

    def counts(self):
        n = self.index.num_taxa
        a, b = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        _check(self.L.cf_counts_get(self.h, a.ctypes.data, b.ctypes.data))
        return a, b

    def reset_counts(self):
        _check(self.L.cf_counts_reset(self.h))

    def counts_single(self):
        """perfect single assignments per taxon of the batches formatted on the device (cf_counts_get_single)"""
        n = self.L.cf_index_num_taxa(self.index.h)
        a = np.zeros(n, dtype=np.uint64)
        _check(self.L.cf_counts_get_single(self.h, a.ctypes.data))
        return a

    def allreduce_counts(self, nccl_comm, stream=None):
        """in-place RCCL sum of the device counters over the communicator's ranks"""
        _check(self.L.cf_counts_allreduce(self.h, nccl_comm, stream))

    def counts_device_ptr(self):
        return self.L.cf_counts_device(self.h)

    def debug_search(self, codes, max_hits=512):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        hf, hr = np.zeros(max_hits, dtype=HIT_DTYPE), np.zeros(max_hits, dtype=HIT_DTYPE)
        n = (C.c_uint32 * 2)()
        _check(self.L.cf_debug_search(self.h, codes.ctypes.data, len(codes), hf.ctypes.data, hr.ctypes.data, max_hits, n))
        return hf[:n[0]], hr[:n[1]]


class Batch:
    def __init__(self, clf, seq, off, seeds, paired=False):
        self.clf, self.L = clf, clf.L
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        self.n_reads = len(off) - 1
        h = C.c_void_p()
        _check(self.L.cf_batch_create(clf.h, seq.ctypes.data, off.ctypes.data, seeds.ctypes.data, self.n_reads,
                                      int(paired), C.byref(h)))
        self.h = h
        self.n_queries = self.L.cf_batch_num_queries(h)

    def close(self):
        if self.h:
            self.L.cf_batch_destroy(self.h)
            self.h = None

    def plan(self, stream=None):
        """Plan + strand records again from the resident reads (cf_batch_plan); returns the device ms."""
        _check(self.L.cf_batch_plan(self.h, stream))
        ms = C.c_float()
        _check(self.L.cf_batch_plan_ms(self.h, C.byref(ms)))
        return ms.value

    def classify(self, stream=None):
        _check(self.L.cf_classify(self.clf.h, self.h, stream))

    def results(self):
        k = self.clf.params.khits
        rows = np.zeros((self.n_queries, k), dtype=ROW_DTYPE)
        n_rows = np.zeros(self.n_queries, dtype=np.uint32)
        score2 = np.zeros(self.n_queries, dtype=np.uint32)
        _check(self.L.cf_batch_results(self.h, rows.ctypes.data, n_rows.ctypes.data, score2.ctypes.data))
        return rows, n_rows, score2

    def results_compact(self):
        """Packed rows (cf_batch_results_compact): rows of query q = rows[first[q] : first[q] + n_rows[q]]."""
        total = C.c_uint64()
        _check(self.L.cf_batch_num_rows(self.h, C.byref(total)))
        rows = np.zeros(total.value, dtype=ROW_DTYPE)
        n_rows = np.zeros(self.n_queries, dtype=np.uint32)
        score2 = np.zeros(self.n_queries, dtype=np.uint32)
        _check(self.L.cf_batch_results_compact(self.h, rows.ctypes.data, total.value, n_rows.ctypes.data, score2.ctypes.data))
        first = np.zeros(self.n_queries + 1, dtype=np.uint64)
        np.cumsum(n_rows, out=first[1:])
        return rows, first, n_rows, score2

    def timings(self):
        ms = (C.c_float * 5)()
        _check(self.L.cf_batch_timings(self.h, C.byref(ms)))
        return list(ms)

    def max_scores(self):
        m = np.zeros(self.n_queries, dtype=np.uint32)
        _check(self.L.cf_batch_max_scores(self.h, m.ctypes.data))
        return m

    def opcounts(self):
        o = OpCounts()
        _check(self.L.cf_batch_opcounts(self.h, C.byref(o)))
        return o


def pack_reads(seq, off):
    """1 byte per base (codes 0..4) + offsets -> the packed form of cf_packed_reads: (bases u64, nmask u32, len u32).
    Read r starts on a 32-base word; base i at bits 2(i%32) of its word i//32; an N has code 0 and its mask bit set."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    ln = (off[1:] - off[:-1]).astype(np.int64)
    words = (ln + 31) >> 5
    woff = np.zeros(len(ln) + 1, dtype=np.int64)
    np.cumsum(words, out=woff[1:])
    nw = int(woff[-1])
    padded = np.zeros(nw * 32, dtype=np.uint8)
    isn = np.zeros(nw * 32, dtype=bool)
    if len(ln) and off[-1] > off[0]:
        # destination of every base: its read's first slot + its index inside the read
        idx = np.arange(int(off[0]), int(off[-1]), dtype=np.int64)
        rd = np.repeat(np.arange(len(ln)), ln)
        dst = woff[:-1][rd] * 32 + (idx - off[:-1].astype(np.int64)[rd])
        b = seq[int(off[0]):int(off[-1])]
        padded[dst] = np.where(b > 3, 0, b)
        isn[dst] = b > 3
    sh = (2 * np.arange(32, dtype=np.uint64))[None, :]
    bases = np.bitwise_or.reduce(padded.reshape(nw, 32).astype(np.uint64) << sh, axis=1) if nw else np.zeros(0, dtype=np.uint64)
    shm = np.arange(32, dtype=np.uint32)[None, :]
    nmask = np.bitwise_or.reduce(isn.reshape(nw, 32).astype(np.uint32) << shm, axis=1) if nw else np.zeros(0, dtype=np.uint32)
    return bases.astype(np.uint64), nmask.astype(np.uint32), ln.astype(np.uint32)


class PinnedArray:
    """a numpy view of pinned host memory (cf_host_alloc)"""

    def __init__(self, L, dtype, n):
        self.L = L
        self.ptr = C.c_void_p()
        dt = np.dtype(dtype)
        _check(L.cf_host_alloc(C.byref(self.ptr), max(1, n) * dt.itemsize))
        self.a = np.frombuffer((C.c_char * (max(1, n) * dt.itemsize)).from_address(self.ptr.value), dtype=dt, count=n)

    def free(self):
        if self.ptr:
            self.a = None
            self.L.cf_host_free(self.ptr)
            self.ptr = None


class Slot:
    """A reusable batch slot driven through the asynchronous ABI: submit(...) returns at once, wait() blocks."""

    def __init__(self, clf, max_reads=0, max_words=0):
        self.clf, self.L = clf, clf.L
        h = C.c_void_p()
        _check(self.L.cf_batch_alloc(clf.h, max_reads, max_words, C.byref(h)))
        self.h = h
        self._keep = None

    def set_limits(self, hit_slots=0, rows_per_pass=0):
        _check(self.L.cf_batch_set_limits(self.h, hit_slots, rows_per_pass))

    def submit(self, bases, nmask, lens, seeds, paired=False, max_len=None, stream=None, streams=None, n_bases=None, nwords=None):
        """the arrays must stay alive (and, for real overlap, be pinned) until wait() returns.  `streams` = (upload,
        kernels, download) HIP streams: the stages chain through events, so copies of one slot overlap kernels of another.
        nmask = None: the N mask in its sparse form, nwords = (word indices u64, mask words u32) (sparse_nmask)"""
        pr = PackedReads()
        pr.bases, pr.len, pr.seeds = bases.ctypes.data, lens.ctypes.data, seeds.ctypes.data
        if nmask is not None:
            pr.nmask = nmask.ctypes.data
        else:
            ni, nm = nwords if nwords is not None else (np.zeros(0, np.uint64), np.zeros(0, np.uint32))
            pr.nmask, pr.n_nwords = None, len(ni)
            if len(ni):
                pr.nword_idx, pr.nword_mask = ni.ctypes.data, nm.ctypes.data
            nmask = (ni, nm)
        pr.n_reads, pr.n_words = len(lens), len(bases)
        pr.n_bases = int(n_bases if n_bases is not None else (lens.sum(dtype=np.uint64) if len(lens) else 0))
        pr.max_len = int(lens.max()) if max_len is None and len(lens) else int(max_len or 0)
        pr.paired = int(paired)
        self._keep = (bases, nmask, lens, seeds, pr)
        if streams is None:
            _check(self.L.cf_batch_submit(self.h, C.byref(pr), stream))
        else:
            _check(self.L.cf_batch_upload_packed_async(self.h, C.byref(pr), streams[0]))
            _check(self.L.cf_classify_async(self.clf.h, self.h, streams[1]))
            _check(self.L.cf_batch_download_async(self.h, streams[2]))

    def set_result_format(self, fmt):
        """RESULTS_ROWS (cf_row + three words per query) or RESULTS_NARROW (16-byte rows + five bytes per query: wait_narrow)"""
        _check(self.L.cf_batch_set_result_format(self.h, int(fmt)))

    def submit_dense(self, bases4, seeds, read_len, paired=False, nwords=None, streams=None, stream=None):
        """reads of ONE length, four bases per byte (dense_pack): 25 bytes per 100-base read across the link instead of 36"""
        dr = DenseReads()
        dr.bases4, dr.seeds, dr.n_reads, dr.read_len, dr.paired = bases4.ctypes.data, seeds.ctypes.data, len(seeds), int(read_len), int(paired)
        ni, nm = nwords if nwords is not None else (np.zeros(0, np.uint64), np.zeros(0, np.uint32))
        dr.n_nwords = len(ni)
        if len(ni):
            dr.nword_idx, dr.nword_mask = ni.ctypes.data, nm.ctypes.data
        self._keep = (bases4, seeds, ni, nm, dr)
        st = streams if streams is not None else (stream, stream, stream)
        _check(self.L.cf_batch_upload_dense_async(self.h, C.byref(dr), st[0]))
        _check(self.L.cf_classify_async(self.clf.h, self.h, st[1]))
        _check(self.L.cf_batch_download_async(self.h, st[2]))

    def wait_narrow(self, copy=True, expand=None):
        """-> rows16, qinfo, score2, info (views of the slot's pinned memory unless copy).  expand = (lens or None, uniform_len,
        paired): also the wide form (cf_results_narrow_expand) -> rows, n_rows, score2, max_score, info"""
        r = ResultsNarrow()
        _check(self.L.cf_batch_wait_narrow(self.h, C.byref(r)))
        nq, tot = r.n_queries, r.total_rows

        def view(ptr, dt, n):
            dt = np.dtype(dt)
            if n == 0:
                return np.zeros(0, dtype=dt)
            a = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt, count=n)
            return a.copy() if copy else a
        info = {"planned_sa_rows": r.planned_sa_rows, "row_passes": r.row_passes, "slow_post": r.slow_post, "slow_score": r.slow_score}
        if expand is None:
            return view(r.rows, ROW16_DTYPE, tot), view(r.qinfo, np.uint8, nq), view(r.score2, np.uint32, nq), info
        lens, uniform, paired = expand
        rows, n_rows, ms = np.zeros(tot, dtype=ROW_DTYPE), np.zeros(nq, dtype=np.uint32), np.zeros(nq, dtype=np.uint32)
        lp = None if lens is None else np.ascontiguousarray(lens, dtype=np.uint32)
        _check(self.L.cf_results_narrow_expand(self.clf.index.h, C.byref(r), None if lp is None else lp.ctypes.data, int(uniform or 0), int(paired),
                                               rows.ctypes.data, n_rows.ctypes.data, ms.ctypes.data))
        return rows, n_rows, view(r.score2, np.uint32, nq), ms, info

    def submit_text(self, text, fmt, seed=0, max_reads=0, stream=None, text2=None):
        """a block of whole FASTA / FASTQ records as the file holds them (cf_batch_upload_text: parsed on the device), then the
        kernels; text2 = the mates' block.  -> TextInfo; info.irregular != 0: the block is not in the plain form, nothing was submitted"""
        buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, dtype=np.uint8)
        tr, info = TextReads(), TextInfo()
        tr.text, tr.n_bytes, tr.format, tr.global_seed, tr.max_reads = buf.ctypes.data, len(text), int(fmt), int(seed), int(max_reads)
        buf2 = None
        if text2 is not None:
            buf2 = np.frombuffer(text2, dtype=np.uint8) if len(text2) else np.zeros(1, dtype=np.uint8)
            tr.text2, tr.n_bytes2 = buf2.ctypes.data, len(text2)
        self._keep = (buf, buf2, tr)
        _check(self.L.cf_batch_upload_text(self.h, C.byref(tr), stream, C.byref(info)))
        if not info.irregular:
            _check(self.L.cf_classify_async(self.clf.h, self.h, stream))
        return info

    def wait_text(self):
        """-> the batch's rows as the default columns' text (bytes), the perfect multi-assignment tuples (u32: n, n taxon indices, ...), info"""
        r = ResultsText()
        _check(self.L.cf_batch_wait_text(self.h, C.byref(r)))
        text = C.string_at(r.text, r.n_bytes) if r.n_bytes else b""
        tuples = np.frombuffer(C.string_at(r.tuples, 4 * r.n_tuple_words), dtype=np.uint32).copy() if r.n_tuple_words else np.zeros(0, np.uint32)
        return text, tuples, {"n_queries": r.n_queries, "total_rows": r.total_rows, "planned_sa_rows": r.planned_sa_rows, "row_passes": r.row_passes,
                              "slow_post": r.slow_post, "slow_score": r.slow_score}

    def resubmit(self, streams):
        """plan + kernels + download once more over the reads the slot holds since its last submit (nothing is uploaded);
        streams = (kernels, download)"""
        _check(self.L.cf_batch_reclassify_async(self.clf.h, self.h, streams[0]))
        _check(self.L.cf_batch_download_async(self.h, streams[1]))

    def plan(self, stream=None):
        """plan + strand records again from the slot's resident reads; returns the device ms"""
        _check(self.L.cf_batch_plan(self.h, stream))
        ms = C.c_float()
        _check(self.L.cf_batch_plan_ms(self.h, C.byref(ms)))
        return ms.value

    def classify(self, stream=None):
        """the kernels again on the slot's resident reads (blocking)"""
        _check(self.L.cf_classify(self.clf.h, self.h, stream))

    def submit_bytes(self, seq, off, seeds, paired=False, stream=None):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        self._keep = (seq, off, seeds)
        _check(self.L.cf_batch_upload(self.h, seq.ctypes.data, off.ctypes.data, seeds.ctypes.data, len(off) - 1, int(paired), stream))
        _check(self.L.cf_classify_async(self.clf.h, self.h, stream))
        _check(self.L.cf_batch_download_async(self.h, stream))

    def wait(self, copy=True, offsets=True):
        """-> rows (packed), first, n_rows, score2, max_score, info dict; views of the slot's pinned memory unless copy.
        offsets=False: `first` (the running sum of n_rows, a host-side pass over every query) is left out (None)"""
        r = Results()
        _check(self.L.cf_batch_wait(self.h, C.byref(r)))
        nq, tot = r.n_queries, r.total_rows

        def view(ptr, dt, n):
            dt = np.dtype(dt)
            if n == 0:
                return np.zeros(0, dtype=dt)
            a = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt, count=n)
            return a.copy() if copy else a
        rows = view(r.rows, ROW_DTYPE, tot)
        n_rows, score2, max_score = view(r.n_rows, np.uint32, nq), view(r.score2, np.uint32, nq), view(r.max_score, np.uint32, nq)
        first = None
        if offsets:
            first = np.zeros(nq + 1, dtype=np.uint64)
            np.cumsum(n_rows, out=first[1:])
        return rows, first, n_rows, score2, max_score, {"planned_sa_rows": r.planned_sa_rows, "row_passes": r.row_passes,
                                                            "slow_post": r.slow_post, "slow_score": r.slow_score}

    def timings(self):
        ms = (C.c_float * 5)()
        _check(self.L.cf_batch_timings(self.h, C.byref(ms)))
        pm = C.c_float()
        _check(self.L.cf_batch_plan_ms(self.h, C.byref(pm)))
        return list(ms), pm.value

    def opcounts(self):
        o = OpCounts()
        _check(self.L.cf_batch_opcounts(self.h, C.byref(o)))
        return o

    def close(self):
        if self.h:
            self.L.cf_batch_destroy(self.h)
            self.h = None


def unpack_rows(rows, first, n_rows, k):
    """packed rows -> the [nq, k] slot layout of Batch.results()"""
    nq = len(n_rows)
    out = np.zeros((nq, k), dtype=ROW_DTYPE)
    if len(rows):
        q = np.repeat(np.arange(nq), n_rows)
        i = np.arange(len(rows)) - np.repeat(first[:-1].astype(np.int64), n_rows)
        out[q, i] = rows
    return out


def build_index(out_base, conversion_table, taxonomy_tree, name_table=None, fasta=None, codes=None, seq_off=None,
                seq_names=None, device=0, off_rate=4, ftab_chars=10, chunk_suffixes=0, verbose=False):
    """GPU index construction (cf_build_index): either `fasta` (list of paths) or in-memory
    `codes` (u8, 0..3 = ACGT, >3 = gap) + `seq_off` (u64, n+1) + `seq_names` (list of bytes).
    Returns the phase timings [parse, gpu, write, total] in seconds."""
    L = lib()
    b = BuildInput()
    _check(L.cf_build_input_default(C.byref(b)))
    keep = []
    if fasta:
        arr = (C.c_char_p * len(fasta))(*[f.encode() for f in fasta])
        keep.append(arr)
        b.fasta_paths, b.n_fasta = arr, len(fasta)
    else:
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        names = (C.c_char_p * len(seq_names))(*seq_names)
        keep += [codes, seq_off, names]
        b.codes, b.seq_off, b.seq_names, b.n_seq = codes.ctypes.data, seq_off.ctypes.data, names, len(seq_names)
    b.conversion_table = conversion_table.encode()
    b.taxonomy_tree = taxonomy_tree.encode()
    b.name_table = name_table.encode() if name_table else None
    b.off_rate, b.ftab_chars, b.chunk_suffixes, b.verbose = off_rate, ftab_chars, chunk_suffixes, int(verbose)
    st = L.cf_build_index(C.byref(b), out_base.encode(), device)
    if st != 0:
        raise CfError("%s: %s" % (L.cf_strerror(st).decode(), L.cf_build_last_error().decode()))
    t = (C.c_double * 4)()
    L.cf_build_timings(C.byref(t))
    return list(t)


def build_taxonomy(out_base, fasta, conversion_table, taxonomy_tree, name_table=None, size_table=None):
    """Host-only half of a build (cf_build_taxonomy): writes <out_base>.3.cf; needs no device."""
    L = lib()
    b = BuildInput()
    _check(L.cf_build_input_default(C.byref(b)))
    arr = (C.c_char_p * len(fasta))(*[f.encode() for f in fasta])
    b.fasta_paths, b.n_fasta = arr, len(fasta)
    b.conversion_table = conversion_table.encode()
    b.taxonomy_tree = taxonomy_tree.encode()
    b.name_table = name_table.encode() if name_table else None
    b.size_table = size_table.encode() if size_table else None
    err = C.create_string_buffer(1024)
    st = L.cf_build_taxonomy(C.byref(b), out_base.encode(), err, len(err))
    if st != 0:
        raise CfError("%s: %s" % (L.cf_strerror(st).decode(), err.value.decode()))


def build_describe(fasta=None, codes=None, seq_off=None, seq_names=None):
    """The builder's bookkeeping of its input (cf_build_describe, no device) — FASTA files, or in-memory
    `codes` / `seq_off` / `seq_names` as for build_index: dict with len, plen, rstarts (n_frag x 3), names,
    text (codes)."""
    import tempfile
    L = lib()
    b = BuildInput()
    _check(L.cf_build_input_default(C.byref(b)))
    if fasta:
        arr = (C.c_char_p * len(fasta))(*[f.encode() for f in fasta])
        b.fasta_paths, b.n_fasta = arr, len(fasta)
    else:
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        arr = (C.c_char_p * len(seq_names))(*seq_names)
        b.codes, b.seq_off, b.seq_names, b.n_seq = codes.ctypes.data, seq_off.ctypes.data, arr, len(seq_names)
    err = C.create_string_buffer(1024)
    with tempfile.TemporaryDirectory() as t:
        path = os.path.join(t, "d.bin")
        st = L.cf_build_describe(C.byref(b), path.encode(), err, len(err))
        if st != 0:
            raise CfError("%s: %s" % (L.cf_strerror(st).decode(), err.value.decode()))
        raw = open(path, "rb").read()
    u = lambda o: int(np.frombuffer(raw, dtype="<u8", count=1, offset=o)[0])   # noqa: E731
    n, npat = u(0), u(8)
    plen = np.frombuffer(raw, dtype="<u8", count=npat, offset=16)
    o = 16 + 8 * npat
    nfrag = u(o)
    rst = np.frombuffer(raw, dtype="<u8", count=3 * nfrag, offset=o + 8).reshape(-1, 3)
    o += 8 + 24 * nfrag
    e = raw.index(b"\0", o)
    names = raw[o:e].split(b"\n")[:-1]
    text = np.frombuffer(raw, dtype=np.uint8, count=n, offset=e + 1)
    return {"len": n, "plen": plen, "rstarts": rst, "names": names, "text": text}


class Report:
    """cf_report: per-taxon counters + observed tuples + EM abundance + report TSV (host side)."""

    def __init__(self, index):
        self.L = index.L
        h = C.c_void_p()
        _check(self.L.cf_report_create(index.h, C.byref(h)))
        self.h = h

    def add(self, rows, n_rows, max_score, khits):
        rows = np.ascontiguousarray(rows)
        n_rows = np.ascontiguousarray(n_rows, dtype=np.uint32)
        max_score = np.ascontiguousarray(max_score, dtype=np.uint32)
        _check(self.L.cf_report_add(self.h, rows.ctypes.data, n_rows.ctypes.data, max_score.ctypes.data, len(n_rows), khits))

    def add_counts(self, taxids, n_reads, n_unique):
        t = np.ascontiguousarray(taxids, dtype=np.uint64)
        a = np.ascontiguousarray(n_reads, dtype=np.uint64)
        b = np.ascontiguousarray(n_unique, dtype=np.uint64)
        _check(self.L.cf_report_add_counts(self.h, t.ctypes.data, a.ctypes.data, b.ctypes.data, len(t)))

    def reset_counts(self):
        _check(self.L.cf_report_reset_counts(self.h))

    def adopt_counts(self, n_reads, n_unique):
        """the devices' per-taxon counters (dense, in the index's taxon order) in place of the tally of the rows: the two must
        agree taxon by taxon, else CfError (the self-check every centrifuge-class run ends with)"""
        a = np.ascontiguousarray(n_reads, dtype=np.uint64)
        b = np.ascontiguousarray(n_unique, dtype=np.uint64)
        _check(self.L.cf_report_adopt_counts(self.h, a.ctypes.data, b.ctypes.data, len(a)))

    def add_tuples(self, tuples):
        t = np.ascontiguousarray(tuples, dtype=np.uint32)
        _check(self.L.cf_report_add_tuples(self.h, t.ctypes.data, len(t)))

    def adopt_device_tally(self, n_reads, n_unique, n_single):
        """the devices' counters as the report's own, the perfect single assignments into the observed tuples (no cross-check)"""
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64) for x in (n_reads, n_unique, n_single))
        _check(self.L.cf_report_adopt_device_tally(self.h, a.ctypes.data, b.ctypes.data, c.ctypes.data, len(a)))

    def serialize(self):
        need = C.c_uint64()
        _check(self.L.cf_report_serialize(self.h, None, 0, C.byref(need)))
        buf = np.zeros(need.value, dtype=np.uint64)
        _check(self.L.cf_report_serialize(self.h, buf.ctypes.data, need.value, C.byref(need)))
        return buf

    def merge(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint64)
        _check(self.L.cf_report_merge(self.h, buf.ctypes.data, len(buf)))

    def write(self, path, abundance=True):
        it, df = C.c_uint64(), C.c_double()
        _check(self.L.cf_report_write(self.h, path.encode(), int(abundance), C.byref(it), C.byref(df)))
        return it.value, df.value

    def close(self):
        if self.h:
            self.L.cf_report_destroy(self.h)
            self.h = None
