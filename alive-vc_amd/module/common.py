"""Drop-in for the inference-time surface of the reference's module/common.py.

  match_features(source, reference, k=4, alpha=0.0)   <- /root/reference/module/common.py:96-109

runs on the MI355X through libalive_vc.so: fp6- (or fp8- / bf16-) MFMA candidate scoring with
LDS-staged top-k' lists, exact fp32 rescoring, gather-mean-blend.  The library
is packed (normalised fp6 / fp8 / bf16 rows + fp32 rows + norms) once per reference tensor
and cached, so the per-window calls of inference.py:129 only pay for the search.
"""
import os
import weakref

import torch

from . import _native as nat

DIM = 768
# Candidate stage of the search: "fp6" (block-scaled e2m3 MFMA at twice the fp8 rate, 96 stationary frames per wave; round 5),
# "fp8" (block-scaled e4m3 MFMA) -- both 32 candidates per frame and library split -- or "bf16" (16 candidates); all of them
# feed the same exact fp32 rescoring and the same per-frame certificate.  env ALIVE_KNN_PREFILTER overrides.
DEFAULT_PREFILTER = "fp6"
_STAGES = ("bf16", "fp8", "fp6")


def _prefilter():
    return os.environ.get("ALIVE_KNN_PREFILTER", DEFAULT_PREFILTER)


def _rotate_allowed():
    """ALIVE_KNN_ROTATE=0 switches the rotated form of the fp8 stage off (dense banks then go through the bf16 stage as in round 5)"""
    return os.environ.get("ALIVE_KNN_ROTATE", "1") not in ("0", "false", "False")


# A bank is searched on ROTATED fp8 operands (csrc/knn.hip rot_codes_kernel, include/alive_vc.h) when it is DENSE -- its rows share a
# common component: the leading eigenvalue of the unit rows' second moment carries at least ROT_MIN_LEAD of the trace -- and lives in a
# subspace of at most 571 dimensions (the reference's ContentEncoder ends in Conv1d(512 -> 768): every bank it produces does).
ROT_MIN_LEAD = 0.10
ROT_MAX_TAIL = 1.0e-9


def _strict():
    """ALIVE_KNN_STRICT=1: every search through the bf16 stage with the DETERMINISTIC certificate (no statistical
    assumption; include/alive_vc.h: alive_knn_search_strict).  The default certificates are statistical (7 sigma of the
    measured stage error; audited in profiles/r03_knn_audit.json)."""
    return os.environ.get("ALIVE_KNN_STRICT", "0") not in ("0", "", "false", "False")


class PackedLibrary:
    """Device-resident search form of a voice library tokens[768, M] (one shard).

    lib_bf16[M_pad,768] normalised rows (MFMA operand), rows[M,768] fp32 raw rows,
    norms[M].  `idx_base` is the global index of row 0 when the library is sharded."""

    def __init__(self, tokens_DxM: torch.Tensor, idx_base: int = 0, prefilter: str = None, strict: bool = None):
        self.strict = _strict() if strict is None else bool(strict)
        self.prefilter = "bf16" if self.strict else (prefilter or _prefilter())
        if self.prefilter not in _STAGES:
            raise ValueError(f"prefilter must be one of {_STAGES}, got {self.prefilter!r}")
        if tokens_DxM.dim() != 2 or tokens_DxM.shape[0] != DIM:
            raise ValueError(f"library must be [768, M], got {tuple(tokens_DxM.shape)}")
        t = tokens_DxM.contiguous().float()
        L = nat.lib()
        self.M = int(t.shape[1])
        self.idx_base = int(idx_base)
        m_pad = L.alive_library_padded_rows(self.M)
        dev = t.device
        self.lib_bf16 = torch.empty(m_pad, DIM, dtype=torch.bfloat16, device=dev)
        self.rows = torch.empty(self.M, DIM, dtype=torch.float32, device=dev)
        self.norms = torch.empty(self.M, dtype=torch.float32, device=dev)
        nat.check(L.alive_library_pack(nat.ptr(t), self.M, DIM, nat.ptr(self.lib_bf16), nat.ptr(self.rows),
                                       nat.ptr(self.norms), nat.stream()), "alive_library_pack")
        # A zero-norm (or non-finite) row: the reference divides it by its norm (common.py:103) and the NaN cosines that
        # follow rank first in every frame's top-k, i.e. the row silently joins every match.  Here it is an error of the
        # library (one reduction + sync per pack, which happens once per bank).
        bad = ~(torch.isfinite(self.norms) & (self.norms > 0))
        if bool(bad.any()):
            first = int(torch.nonzero(bad)[0])
            raise ValueError(f"voice library row {first + self.idx_base} has zero or non-finite norm ({int(bad.sum())} such rows): "
                             f"remove them (the reference would match every frame to them through NaN cosines)")
        self.lib_f8 = None                 # the fp8 OR the fp6 image of the rows (same size and tile layout), per self.prefilter
        self.fp6_declined = False
        self.rot = None                    # rotated fp8 stage: {"W": packed 1x1 conv weights of the basis, "spectrum": ...}
        if self.prefilter in ("fp6", "fp8") and self.M >= 4096 and _rotate_allowed():
            self._try_rotated()            # a dense bank: the fp8 stage on rotated operands (sets self.rot, self.lib_f8, prefilter "fp8")
        if self.rot is None and self.prefilter == "fp6" and self.M > 0:
            # e2m3 x 2^5 ends at 7.5 / 32 = 0.234: a row of a unit-norm library with a larger element (a few dominant coordinates:
            # "spiky" banks, never a dense content-encoder bank, whose elements are ~0.036 +- a few sigma) would be CLIPPED, and a clipped
            # row's score is low by more than any per-frame error statistic shows -- such a library is searched through the fp8 stage
            # (e4m3 x 2^8 reaches 1.75).  One reduction + sync per pack.  (A frame that clips is caught per search: knn.hip force_fail.)
            if self._fp6_would_clip():
                self.prefilter, self.fp6_declined = "fp8", True
        if self.rot is None and self.prefilter in ("fp8", "fp6"):
            self.lib_f8 = self._pack_stage(self.prefilter)
        self.bound = None
        self.lib_lo = None
        if self.strict:
            self._make_bound()
        self._ws = nat.Workspace()

    def _fp6_would_clip(self):
        if self.M == 0:
            return False
        lo, hi = torch.aminmax(self.lib_bf16[:self.M])        # (no |x| temporary: the image is 1.5 GB at 1 M rows)
        return max(-float(lo), float(hi)) > 7.75 / 32.0

    def _pack_stage(self, prefilter):
        L = nat.lib()
        buf = torch.empty(L.alive_library_fp8_bytes(self.M), dtype=torch.uint8, device=self.rows.device)
        fn = L.alive_library_pack_fp6 if prefilter == "fp6" else L.alive_library_pack_fp8
        nat.check(fn(nat.ptr(self.lib_bf16), self.M, nat.ptr(buf), nat.stream()), "alive_library_pack_" + prefilter)
        return buf

    def _try_rotated(self, chunk=131072):
        """Decide whether this bank is searched on rotated fp8 operands and, if so, build the basis and the rows' codes (once per bank:
        torch is used for the 768 x 768 second moment, its eigen-decomposition and the rows' change of basis -- pack-time plumbing; the
        codes, the frames' change of basis (alive_conv1d) and the search are the library's own kernels)."""
        from ._pack import pack_conv_split
        L = nat.lib()
        rc, ra, rm = L.alive_knn_rot_coordinates(), L.alive_knn_rot_leading(), L.alive_knn_rot_mixed()
        used = ra + rm                                          # 571 directions carry the bank; the 5 coordinates behind them are not read
        dev = self.rows.device
        C = torch.zeros(DIM, DIM, dtype=torch.float64, device=dev)
        mean = torch.zeros(DIM, dtype=torch.float64, device=dev)
        for s0 in range(0, self.M, chunk):
            ln = self.rows[s0:s0 + chunk] / self.norms[s0:s0 + chunk, None]
            C += (ln.t() @ ln).double()
            mean += ln.double().sum(0)
        evals, evecs = torch.linalg.eigh((C / self.M).cpu())
        order = torch.argsort(evals, descending=True)
        evals, U = evals[order].clamp(min=0.0), evecs[:, order]
        total = float(evals.sum())
        lead, tail = float(evals[0]) / total, float(evals[used:].sum()) / total
        self.rot_spectrum = {"leading_eigenvalue_share": lead, "energy_beyond_571_directions": tail}
        if not (lead >= ROT_MIN_LEAD and tail <= ROT_MAX_TAIL):
            return
        a0 = float((mean.cpu() / self.M) @ U[:, 0])             # mean of the rows' leading coordinate: make it positive
        if a0 < 0:
            U[:, 0], a0 = -U[:, 0], -a0
        c0 = float((torch.tensor([a0 * 256.0]).to(torch.float8_e4m3fn).float() / 256.0)[0])      # e4m3-exact centring constant
        g = torch.Generator().manual_seed(20261004)
        R = torch.linalg.qr(torch.randn(rm, rm, generator=g, dtype=torch.float64))[0]
        W = torch.zeros(DIM, rc, dtype=torch.float64)
        W[:, :ra] = U[:, :ra]
        W[:, ra:used] = U[:, ra:used] @ R
        W = W.float().to(dev)                                   # [768, 576]: 571 orthonormal columns, 5 zero ones
        buf = torch.empty(L.alive_library_fp8_bytes(self.M), dtype=torch.uint8, device=dev)
        for s0 in range(0, self.M, chunk):                                                    # (chunk is a multiple of 32)
            ln = self.rows[s0:s0 + chunk] / self.norms[s0:s0 + chunk, None]
            y = (ln @ W).contiguous()
            nat.check(L.alive_library_pack_fp8_rot(nat.ptr(y), s0, y.shape[0], self.M, c0, nat.ptr(buf), nat.stream()), "alive_library_pack_fp8_rot")
        self.lib_f8 = buf
        self.prefilter = "fp8"
        # the frames' change of basis is a 1x1 conv 768 -> 576 on two bf16 planes (2^-16: far below the stage's 8-bit digits)
        self.rot = {"W": pack_conv_split(W.t().contiguous().unsqueeze(2), 2), "co": rc, "basis": W, "c0": c0}

    def _rotate_frames(self, source):
        """source [n, 768, t] -> W^T source [n, 576, t] through alive_conv1d (split bf16)"""
        import ctypes as C
        n, d, t = source.shape
        y = torch.empty(n, self.rot["co"], t, device=source.device)
        W = self.rot["W"]
        dsc = nat.AliveConv()
        dsc.W, dsc.bias, dsc.X = nat.ptr(W), None, nat.ptr(source)
        dsc.N, dsc.Ci, dsc.Tin, dsc.Co, dsc.K_pad = n, d, t, self.rot["co"], W.shape[1] * 32
        dsc.precision, dsc.Ci_pad = 1, d
        dsc.KW, dsc.stride, dsc.dil, dsc.pad_left, dsc.pad_mode = 1, 1, 1, 0, 0
        dsc.Tout, dsc.up, dsc.act = t, 1, 0
        dsc.Y = nat.ptr(y)
        nat.check(nat.lib().alive_conv1d(C.byref(dsc), nat.stream()), "alive_conv1d (frames -> the bank's basis)")
        return y

    def _make_bound(self):
        """max_R || r^ - bf16(r^) ||: the library's share of the strict certificate's deterministic bound -- and the library's lo
        plane bf16(r^ - bf16(r^)) for the split-bf16 collect tier behind it (ALIVE_KNN_STRICT_SPLIT=0: without, the round-3 form)"""
        self.bound = torch.zeros(1, dtype=torch.float32, device=self.rows.device)
        nat.check(nat.lib().alive_library_rounding_bound(nat.ptr(self.lib_bf16), nat.ptr(self.rows), nat.ptr(self.norms), self.M,
                                                         nat.ptr(self.bound), nat.stream()), "alive_library_rounding_bound")
        self.lib_lo = None
        if os.environ.get("ALIVE_KNN_STRICT_SPLIT", "1") not in ("0", "false", "False"):
            self.lib_lo = torch.empty_like(self.lib_bf16)
            nat.check(nat.lib().alive_library_pack_lo(nat.ptr(self.lib_bf16), nat.ptr(self.rows), nat.ptr(self.norms), self.M,
                                                      nat.ptr(self.lib_lo), nat.stream()), "alive_library_pack_lo")

    def search(self, source, k, events=None):
        """exact top-k of this shard: (val[Tt,k] fp32 desc, idx[Tt,k] int32 global).
        events = (start, stop): torch.cuda.Event pair recorded around the first-stage scoring kernel (measurement)."""
        n, d, t = source.shape
        L = nat.lib()
        val = torch.empty(n * t, k, dtype=torch.float32, device=source.device)
        idx = torch.empty(n * t, k, dtype=torch.int32, device=source.device)
        split = self.strict and self.lib_lo is not None          # the split-bf16 collect tier needs the frames' two planes as well
        ws = self._ws.get((L.alive_knn_workspace_bytes_strict if split else L.alive_knn_workspace_bytes_fast)(n * t, self.M), source.device)
        ev = (None, None) if events is None else (events[0].cuda_event, events[1].cuda_event)
        if self.strict:
            nat.check(L.alive_knn_search_strict(nat.ptr(source), n, t, nat.ptr(self.lib_bf16), nat.ptr(self.lib_lo), nat.ptr(self.rows), nat.ptr(self.norms),
                                                nat.ptr(self.bound), self.M, self.idx_base, k, nat.ptr(val), nat.ptr(idx),
                                                nat.ptr(ws), nat.stream(), *ev), "alive_knn_search_strict")
        elif self.rot is not None:
            y = self._rotate_frames(source)
            nat.check(L.alive_knn_search_fp8_rot_timed(nat.ptr(source), nat.ptr(y), self.rot["c0"], n, t, nat.ptr(self.lib_f8), nat.ptr(self.lib_bf16),
                                                       nat.ptr(self.rows), nat.ptr(self.norms), self.M, self.idx_base, k,
                                                       nat.ptr(val), nat.ptr(idx), nat.ptr(ws), nat.stream(), *ev), "alive_knn_search_fp8_rot")
        elif self.lib_f8 is not None:
            fn = L.alive_knn_search_fp6_timed if self.prefilter == "fp6" else L.alive_knn_search_fp8_timed
            nat.check(fn(nat.ptr(source), n, t, nat.ptr(self.lib_f8), nat.ptr(self.lib_bf16),
                         nat.ptr(self.rows), nat.ptr(self.norms), self.M, self.idx_base, k,
                         nat.ptr(val), nat.ptr(idx), nat.ptr(ws), nat.stream(), *ev), "alive_knn_search_" + self.prefilter)
        else:
            nat.check(L.alive_knn_search_timed(nat.ptr(source), n, t, nat.ptr(self.lib_bf16), nat.ptr(self.rows),
                                               nat.ptr(self.norms), self.M, self.idx_base, k, nat.ptr(val), nat.ptr(idx),
                                               nat.ptr(ws), nat.stream(), *ev), "alive_knn_search")
        self._last = (n, t, k, ws)
        return val, idx


    def with_prefilter(self, prefilter, rotated=False):
        """the same resident library searched through the other candidate stage (shares every tensor); the PLAIN form of that stage
        unless rotated=True and this library has a rotated fp8 image"""
        import copy
        if prefilter not in _STAGES:
            raise ValueError(prefilter)
        other = copy.copy(self)
        other.__dict__.pop("search", None)             # an instrumented search (bench.py) stays with the original
        if rotated and self.rot is not None and prefilter == "fp8":
            other._ws, other._last = nat.Workspace(), None
            other.strict, other.bound, other.lib_lo = False, None, None
            return other
        was_rot, other.rot = self.rot is not None, None
        other.fp6_declined = prefilter == "fp6" and self._fp6_would_clip()
        if other.fp6_declined:
            prefilter = "fp8"
        other.prefilter = prefilter
        other.strict, other.bound, other.lib_lo = False, None, None
        other._ws = nat.Workspace()
        other._last = None
        if prefilter == "bf16":
            other.lib_f8 = None
        elif self.lib_f8 is None or self.prefilter != prefilter or was_rot:
            other.lib_f8 = self._pack_stage(prefilter)
        return other

    def with_strict(self):
        """the same resident library searched with the deterministic certificate (shares every tensor)"""
        other = self.with_prefilter("bf16")
        other.strict = True
        other._make_bound()
        return other

    def search_stats(self):
        """what the tiers of the last search on the current stream did (syncs; tests / bench)"""
        last = getattr(self, "_last", None)
        if last is None:
            return None
        n, t, k, ws = last
        torch.cuda.synchronize()
        st = {"prefilter": self.prefilter, "certificate": "deterministic" if self.strict else "statistical"}
        if getattr(self, "rot", None) is not None:
            st["rotated_operands"] = True          # dense bank: the fp8 stage on the bank's own basis (csrc/knn.hip rot_codes_kernel)
        off = nat.lib().alive_knn_search_stats(n, t, self.M, nat.ptr(ws)) - ws.data_ptr()
        c = ws[off:off + 64].view(torch.int32).tolist()       # int[16]: knn.hip ST_*
        tier = c[7]                                     # written by the C side on every path (knn.hip: ST_TIER)
        if tier == 1:
            return dict(st, tier="exact scan of every row (streaming)")
        if tier == 2:
            return dict(st, tier="exact scan of every row (k > 8)")
        if tier not in (3, 4, 5):
            raise RuntimeError(f"alive_knn_search_stats: no search has run on this workspace (tier word {tier})")
        few = 0
        if tier in (4, 5):              # fp8 / fp6 stage first (the counters keep their fp8 names: "the first, low-precision stage")
            # [0] frames that failed the fp8 certificate: up to [11] of them go straight to the exact scan (knn.hip RESEARCH_MIN)
            few = c[0] if c[0] <= c[11] else 0
            st.update(frames_failed_fp8_certificate=c[0], frames_researched_on_bf16=0 if few else c[0], probe_sample=c[2],
                      probe_failed_fp8_certificate=c[3], probe_chose_bf16_first=bool(c[4]), fp8_blocks_seeded=c[9],
                      probe_skipped_on_history=bool(c[13]))      # knn.hip ST_PROBE_SKIPPED: the workspace's last searches all went one way
        # [1] frames that failed the bf16 certificate: up to [12] of them go straight to the exact scan (knn.hip COLLECT_MIN),
        # more go through the collect tier and only its overflow ([8]) is scanned exactly
        direct = c[1] <= c[12]
        split = self.strict and getattr(self, "lib_lo", None) is not None       # the collect tier is the split-bf16 pass then
        st.update(bf16_blocks_seeded=c[10], frames_failed_bf16_certificate=c[1],
                  frames_collected_on_bf16=0 if direct or split else c[1],
                  frames_searched_exactly=few + (c[1] if direct else c[8]), frames=n * t)
        if split:
            st.update(frames_collected_on_split_bf16=0 if direct else c[1])
        return st

    def fallback_frames(self):
        """frames the last fp8 search had to search again, on bf16 or exactly (syncs; tests / bench)"""
        st = self.search_stats()
        return 0 if st is None else int(st.get("frames_failed_fp8_certificate") or 0)


def merge_gather(cand_val, cand_idx, n_shards, k, alpha, rows_full, source, return_indices=False):
    """merge [S, Tt, k] exact lists, gather rows, mean over k, alpha-blend (common.py:107-109)."""
    n, d, t = source.shape
    out = torch.empty_like(source)
    fin = torch.empty(n * t, k, dtype=torch.int32, device=source.device) if return_indices else None
    nat.check(nat.lib().alive_knn_merge_gather(nat.ptr(cand_val), nat.ptr(cand_idx), n_shards, k, float(alpha),
                                               nat.ptr(rows_full), nat.ptr(source), n, t, nat.ptr(out), nat.ptr(fin),
                                               nat.stream()), "alive_knn_merge_gather")
    return (out, fin) if return_indices else out


_cache = {}          # key -> (PackedLibrary, weakref to the tensor that owns the storage)


def _packed_for(reference_DxM):
    """Packed form of a library tensor, cached per source tensor.  The key alone (address, shape, version) cannot tell
    a library from a later one of the same shape that the caching allocator put at the same address, so an entry also
    holds a weak reference to the tensor that owns the storage: it is valid only while that very tensor is alive and
    is dropped the moment it dies.  (Writes that bypass autograd's version counter -- ctypes kernels, `.data` views --
    are invisible to it: call `forget_packed()` after such a write.)"""
    owner = reference_DxM._base if reference_DxM._base is not None else reference_DxM
    key = (reference_DxM.data_ptr(), tuple(reference_DxM.shape), tuple(reference_DxM.stride()), reference_DxM._version,
           str(reference_DxM.device), _prefilter(), _strict())
    hit = _cache.get(key)
    if hit is not None and hit[1]() is owner:
        return hit[0]
    if len(_cache) >= 4:
        _cache.pop(next(iter(_cache)))
    packed = PackedLibrary(reference_DxM)
    _cache[key] = (packed, weakref.ref(owner, lambda _r, key=key: _cache.pop(key, None)))
    return packed


def forget_packed():
    """drop every cached packed library (after writing into a library tensor behind torch's back)"""
    _cache.clear()


def match_features(source, reference, k=4, alpha=0.0, return_indices=False):
    """source [N,768,T], reference [N or 1,768,M] -> [N,768,T].  Same contract as the reference function;
    shape errors surface as ValueError (the reference raises RuntimeError from torch.bmm / topk)."""
    if source.dim() != 3 or reference.dim() != 3 or source.shape[1] != DIM or reference.shape[1] != DIM:
        raise ValueError("match_features expects source [N,768,T] and reference [N,768,M]")
    if reference.shape[0] not in (1, source.shape[0]):
        raise ValueError("match_features: batch of reference must be 1 or equal to the batch of source")
    if reference.shape[2] < k:
        raise ValueError(f"match_features: library has {reference.shape[2]} vectors, fewer than k={k}")
    source = source.contiguous().float()
    if reference.shape[0] == 1:
        lib = _packed_for(reference[0])
        val, idx = lib.search(source, k)
        return merge_gather(val, idx, 1, k, alpha, lib.rows, source, return_indices)
    outs, idxs = [], []
    for n in range(source.shape[0]):
        lib = _packed_for(reference[n])
        s = source[n:n + 1].contiguous()
        val, idx = lib.search(s, k)
        r = merge_gather(val, idx, 1, k, alpha, lib.rows, s, return_indices)
        outs.append(r[0] if return_indices else r)
        if return_indices:
            idxs.append(r[1])
    out = torch.cat(outs, 0)
    return (out, torch.cat(idxs, 0)) if return_indices else out
