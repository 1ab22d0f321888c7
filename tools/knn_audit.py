"""Exactness audit of the tiered kNN search: every frame of a batch against a brute-force fp32
scan of the whole library, on the bench's libraries and on adversarial ones, for every first stage; plus the empirical
distribution of the candidate stages' score errors.

  python tools/knn_audit.py audit  [--M 1000000] [--cases randn,dense,...] [--modes fp8,bf16,strict] [--out file.json]
  python tools/knn_audit.py zstats [--M 1000000] [--frames 128] [--out file.json]

A frame counts as a MISMATCH when its returned index set differs from the brute-force top-k set although the brute-force
gap(k, k+1) is >= 1e-5 (below that the fp32 reference itself is order-ambiguous).  Any mismatch is a
false certification (or a broken tier) and a red test (tests/test_gpu_knn_audit.py runs reduced forms of the same cases).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd"))
sys.path.insert(0, ROOT)

DIM = 768


# ------------------------------------------------------------------------------------------- brute force
def normalise_rows(x):
    """[n, 768] -> unit rows, the reference's arithmetic (divide by the L2 norm, common.py:100-103)"""
    return x / x.norm(dim=1, keepdim=True)


def brute_topk(qn, ln, k1, fchunk=8192, lchunk=125_000):
    """exact fp32 top-k1 of qn[T,768] . ln[M,768]^T by a chunked GEMM (never more than fchunk x lchunk scores alive).
    returns (val[T,k1] descending, idx[T,k1] int64)"""
    T, M = qn.shape[0], ln.shape[0]
    k1 = min(k1, M)
    vals = torch.empty(T, k1, device=qn.device)
    idxs = torch.empty(T, k1, dtype=torch.long, device=qn.device)
    lt = ln.t().contiguous()
    for f0 in range(0, T, fchunk):
        q = qn[f0:f0 + fchunk]
        bv = torch.full((q.shape[0], k1), -float("inf"), device=qn.device)
        bi = torch.zeros(q.shape[0], k1, dtype=torch.long, device=qn.device)
        for c in range(0, M, lchunk):
            sc = q @ lt[:, c:c + lchunk]
            v, i = torch.topk(sc, min(k1, sc.shape[1]), dim=1)
            av, ai = torch.cat([bv, v], 1), torch.cat([bi, i + c], 1)
            bv, o = torch.topk(av, k1, dim=1)
            bi = torch.gather(ai, 1, o)
            del sc
        vals[f0:f0 + fchunk], idxs[f0:f0 + fchunk] = bv, bi
    return vals, idxs


def compare(val, idx, bv, bi, k, gap=1e-5):
    """search result (val, idx [T,k]) against brute force (bv, bi [T,k+1]) -> counters"""
    if bv.shape[1] > k:
        safe = (bv[:, k - 1] - bv[:, k]) > gap
    else:
        safe = torch.ones(bv.shape[0], dtype=torch.bool, device=bv.device)
    got = torch.sort(idx.long(), 1).values
    want = torch.sort(bi[:, :k], 1).values
    wrong = (got != want).any(1)
    # a returned set is also acceptable when its k-th exact value equals the brute-force k-th value to fp32 rounding
    # (ties / duplicate rows: a different but equally good row)
    tie_ok = (val[:, k - 1] - bv[:, k - 1]).abs() <= 2e-6
    bad = wrong & safe & ~tie_ok
    return {"frames": int(val.shape[0]), "safe_frames": int(safe.sum()), "mismatches": int(bad.sum()),
            "mismatches_ignoring_value_ties": int((wrong & safe).sum()),
            "max_abs_value_error": float((val - bv[:, :k]).abs().max()),
            "first_bad_frames": bad.nonzero().flatten()[:8].tolist()}


# ------------------------------------------------------------------------------------------- libraries and queries
def gen(dev, seed):
    return torch.Generator(device=dev).manual_seed(seed)


def spiky_rows(M, dev, seed, noise=0.02):
    """rows with 1..8 dominant coordinates (+-U(0.5, 1)) over a small dense floor"""
    g = gen(dev, seed)
    rows = noise * torch.randn(M, DIM, device=dev, generator=g)
    nd = torch.randint(1, 9, (M,), device=dev, generator=g)
    for j in range(8):
        pos = torch.randint(0, DIM, (M,), device=dev, generator=g)
        amp = (0.5 + 0.5 * torch.rand(M, device=dev, generator=g)) * (torch.randint(0, 2, (M,), device=dev, generator=g) * 2 - 1)
        amp = torch.where(nd > j, amp, torch.zeros_like(amp))
        rows[torch.arange(M, device=dev), pos] += amp
    return rows


def make_case(name, M, T, dev, conv=None, bench=None):
    """-> (tokens[768, M], feat[N, 768, t]) of an audit case; T = number of query frames wanted (rounded to windows of 450)"""
    nwin = max(1, T // 450)
    g = gen(dev, 4242)
    if name == "randn":                     # the bench's library and the bench's batch
        toks = torch.randn(DIM, M, device=dev, generator=gen(dev, 1234))
        return toks, bench_features(conv, bench, dev, nwin)
    if name == "randn_iid":                 # i.i.d. Gaussian queries (the scoring kernel's "uncorrelated" leg)
        toks = torch.randn(DIM, M, device=dev, generator=gen(dev, 1234))
        return toks, torch.randn(nwin, DIM, 450, device=dev, generator=gen(dev, 77))
    if name == "dense":                     # CE-derived library, the bench's batch (bench leg `clustered_library`)
        return bench.ce_derived_tokens(conv, M, dev), bench_features(conv, bench, dev, nwin)
    if name == "spiky":                     # spiky rows, dense queries
        rows = spiky_rows(M, dev, 11)
        return rows.t().contiguous(), torch.randn(nwin, DIM, 450, device=dev, generator=g)
    if name == "spiky_self":                # spiky rows, queries = noisy copies of library rows
        rows = spiky_rows(M, dev, 11)
        pick = torch.randint(0, M, (nwin * 450,), device=dev, generator=g)
        q = rows[pick] + 0.05 * torch.randn(nwin * 450, DIM, device=dev, generator=g)
        return rows.t().contiguous(), q.view(nwin, 450, DIM).permute(0, 2, 1).contiguous()
    if name == "norms":                     # row and query norms spanning 1e-3 .. 1e3
        rows = torch.randn(M, DIM, device=dev, generator=gen(dev, 1234))
        rows = rows * torch.pow(10.0, 6 * torch.rand(M, 1, device=dev, generator=g) - 3)
        q = torch.randn(nwin, DIM, 450, device=dev, generator=g)
        q = q * torch.pow(10.0, 6 * torch.rand(nwin, 1, 450, device=dev, generator=g) - 3)
        return rows.t().contiguous(), q
    if name == "mixture":                   # dense rows + a tenth spiky rows + every pure axis row; queries = dense + one spike
        rows = torch.randn(M, DIM, device=dev, generator=gen(dev, 1234))
        ns = M // 10
        rows[:ns] = spiky_rows(ns, dev, 12) * 27.7          # at the dense rows' norm
        rows[ns:ns + DIM] = torch.eye(DIM, device=dev) * 27.7
        # the spike is sized so that q_j * r_j lands where the dense rows' best cosines lie (~0.17 at 1 M rows): the true
        # neighbours then mix rows whose fp8 error is one large term with rows whose error is a sum of 768 small ones
        q = torch.randn(nwin * 450, DIM, device=dev, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        pos = torch.randint(0, DIM, (nwin * 450,), device=dev, generator=g)
        amp = 0.10 + 0.25 * torch.rand(nwin * 450, device=dev, generator=g)
        q[torch.arange(nwin * 450, device=dev), pos] += amp
        return rows.t().contiguous(), q.view(nwin, 450, DIM).permute(0, 2, 1).contiguous()
    if name == "lowrank":                   # rank-16 rows; queries half inside the subspace, half generic
        B = torch.linalg.qr(torch.randn(DIM, 16, device=dev, generator=g))[0]            # [768, 16]
        rows = torch.randn(M, 16, device=dev, generator=gen(dev, 1234)) @ B.t()
        qi = torch.randn(nwin * 450, 16, device=dev, generator=g) @ B.t()
        qo = torch.randn(nwin * 450, DIM, device=dev, generator=g) / 27.7
        half = (torch.arange(nwin * 450, device=dev) % 2 == 0).unsqueeze(1)
        q = torch.where(half, qi, qi * 0.3 + qo)
        return rows.t().contiguous(), q.view(nwin, 450, DIM).permute(0, 2, 1).contiguous()
    if name == "self":                      # queries are exact library rows (randn library)
        toks = torch.randn(DIM, M, device=dev, generator=gen(dev, 1234))
        pick = torch.randint(0, M, (nwin * 450,), device=dev, generator=g)
        q = toks[:, pick].t().contiguous()
        return toks, q.view(nwin, 450, DIM).permute(0, 2, 1).contiguous()
    if name == "dense_self":                # queries are exact rows of the dense library
        toks = bench.ce_derived_tokens(conv, M, dev)
        pick = torch.randint(0, M, (nwin * 450,), device=dev, generator=g)
        q = toks[:, pick].t().contiguous()
        return toks, q.view(nwin, 450, DIM).permute(0, 2, 1).contiguous()
    if name == "clusters":                  # clusters of 40 near-copies: more rows inside any stage's error than a list holds
        nc = M // 40
        cen = torch.randn(nc, DIM, device=dev, generator=gen(dev, 1234))
        rows = cen.repeat_interleave(40, 0)[:M] + 0.05 * torch.randn(M, DIM, device=dev, generator=g)
        pick = torch.randint(0, nc, (nwin * 450,), device=dev, generator=g)
        q = cen[pick] + 0.05 * torch.randn(nwin * 450, DIM, device=dev, generator=g)
        return rows.t().contiguous(), q.view(nwin, 450, DIM).permute(0, 2, 1).contiguous()
    raise ValueError(name)


_feat_cache = {}


def bench_features(conv, bench, dev, nwin):
    """content features of the bench batch's windows (64 utterances x 10 s -> 384 windows), first nwin windows"""
    if "f" not in _feat_cache:
        windows = bench.synth_windows(64, 10.0, 48000, dev, seed=100)
        _feat_cache["f"] = torch.cat([conv.features(windows[i:i + 128])[0] for i in range(0, windows.shape[0], 128)], 0)
    return _feat_cache["f"][:nwin].contiguous()


ALL_CASES = ("randn", "randn_iid", "dense", "spiky", "spiky_self", "norms", "mixture", "lowrank", "self", "dense_self", "clusters")
FULL_BATCH = ("randn", "dense")           # audited on all 172 800 frames of the bench batch; the others on --frames


def run_audit(args):
    import bench
    from module.common import PackedLibrary
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.pipeline import Converter
    dev = torch.device("cuda")
    conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev)
    report = {"M": args.M, "k": args.k, "cases": {}}
    cases = ALL_CASES if args.cases == "all" else tuple(args.cases.split(","))
    modes = tuple(args.modes.split(","))
    for name in cases:
        T = 172_800 if name in FULL_BATCH and not args.quick else args.frames
        t0 = time.time()
        toks, feat = make_case(name, args.M, T, dev, conv, bench)
        N, _, t = feat.shape
        qn = normalise_rows(feat.permute(0, 2, 1).reshape(N * t, DIM))
        ln = normalise_rows(toks.t().contiguous())
        bv, bi = brute_topk(qn, ln, args.k + 1)
        del ln
        torch.cuda.synchronize()
        entry = {"frames": N * t, "brute_force_s": round(time.time() - t0, 1),
                 "gap_k_k1_quantiles": [round(float(x), 6) for x in torch.quantile((bv[:, args.k - 1] - bv[:, args.k]).float(),
                                                                                    torch.tensor([0.001, 0.01, 0.5], device=dev))]}
        for mode in modes:
            strict = mode.startswith("strict")
            pf = "bf16" if mode in ("bf16", "strict") else "fp8"
            lib = PackedLibrary(toks, prefilter=pf, **({"strict": True} if strict else {}))
            lib.search(feat, args.k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            val, idx = lib.search(feat, args.k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) * 1e3
            r = compare(val, idx, bv, bi, args.k)
            r["search_ms"] = round(ms, 2)
            r["tiers"] = lib.search_stats()
            entry[mode] = r
            print(name, mode, json.dumps(r), flush=True)
            del lib
        report["cases"][name] = entry
        del toks, feat, qn, bv, bi
        torch.cuda.empty_cache()
    report["total_mismatches"] = sum(v[m]["mismatches"] for v in report["cases"].values() for m in modes)
    print(json.dumps(report), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)
    return report


# ------------------------------------------------------------------------------------------- error statistics
def stage_operands(x_unit, fmt):
    """what a candidate stage multiplies: bf16(x) or e4m3(bf16(x) * 256) / 256, as fp32 (knn.hip: src_prep / lib_pack,
    lib_to_fp8 / src_to_fp8)"""
    b = x_unit.bfloat16().float()
    if fmt == "bf16":
        return b
    return (b * 256.0).to(torch.float8_e4m3fn).float() / 256.0


def moments(z):
    z = z.double()
    m, s = z.mean(), z.std()
    c = (z - m) / s
    return {"n": int(z.numel()), "mean": float(m), "std": float(s), "skew": float((c ** 3).mean()),
            "excess_kurtosis": float((c ** 4).mean() - 3.0), "max_abs_over_std": float(c.abs().max()),
            "tail_beyond": {str(t): float((c.abs() > t).double().mean()) for t in (4, 5, 6, 7)}}


def emulate_certificate(pre, exact, fmt, M, k=4, z=7.0, splits=3):
    """what the tiered search's first certificate would say for these frames, from full score matrices: the lists are the
    kernel's (library cut into `splits` ranges x the two row groups of a half-wave, (row % 8) // 4), depth 16 (fp8) or 8
    (bf16) per list; c = best list floor; sigma = RMS stage error of the frame's 64 best stage scores (floored by the
    stage prior); certified iff v_k(exact, among the candidates) - c > z sigma.  Also: how many rows a threshold pass
    `stage score >= v_k - z sigma` would collect per frame (the collect tier's buffer)."""
    depth = 16 if fmt == "fp8" else 8
    prior = 1.5e-3 if fmt == "fp8" else 8e-5
    F = pre.shape[0]
    rows = torch.arange(M, device=pre.device)
    per = (M + splits - 1) // splits
    sub = (rows // per) * 2 + (rows % 8) // 4
    c = torch.full((F,), -float("inf"), device=pre.device)
    cand = []
    for sidx in range(2 * splits):
        m = (sub == sidx).nonzero().flatten()
        v, i = torch.topk(pre[:, m], depth, dim=1)
        c = torch.maximum(c, v[:, -1])
        cand.append(m[i])
    cand = torch.cat(cand, 1)                                            # [F, 2 * splits * depth]
    ce, cp = torch.gather(exact, 1, cand), torch.gather(pre, 1, cand)
    top = torch.topk(cp, min(64, cp.shape[1]), dim=1).indices
    err = torch.gather(cp - ce, 1, top)
    sig = torch.clamp(err.pow(2).mean(1).sqrt(), min=prior)
    vk = torch.topk(ce, k, dim=1).values[:, -1]
    true_vk = torch.topk(exact, k, dim=1).values[:, -1]
    passed = (vk - c) > z * sig
    wrong = passed & (vk < true_vk - 1e-6)
    out = {"frames": F, "pass_fraction": float(passed.float().mean()), "false_certifications": int(wrong.sum()),
           "margin_vk_minus_c_quantiles": [round(float(x), 5) for x in torch.quantile(vk - c, torch.tensor([0.01, 0.1, 0.5, 0.9], device=pre.device))],
           "sigma_quantiles": [round(float(x), 6) for x in torch.quantile(sig, torch.tensor([0.1, 0.5, 0.9], device=pre.device))]}
    for zz in (5.0, 6.0, 7.0):
        n = (pre >= (vk - zz * sig).unsqueeze(1)).sum(1).float()
        out[f"rows_collected_z{zz:g}_quantiles"] = [float(x) for x in torch.quantile(n, torch.tensor([0.5, 0.9, 0.99, 1.0], device=pre.device))]
    return out


def run_zstats(args):
    """distribution of (stage score - exact cosine) over frames x all library rows, per format and library family; and the
    same restricted to each frame's 64 best stage scores (what a certificate that estimates the error from its own
    candidates sees: a selected sample)"""
    import bench
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.pipeline import Converter
    dev = torch.device("cuda")
    conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev)
    report = {"M": args.M, "frames": args.frames, "pairs_per_case": args.M * args.frames, "cases": {}}
    for name in args.cases.split(","):
        toks, feat = make_case(name, args.M, max(args.frames, 450 * 384 if name in FULL_BATCH else args.frames), dev, conv, bench)
        N, _, t = feat.shape
        flat = feat.permute(0, 2, 1).reshape(N * t, DIM)
        sel = torch.randperm(N * t, device=dev, generator=gen(dev, 5))[:args.frames]
        qn = normalise_rows(flat[sel])
        ln = normalise_rows(toks.t().contiguous())
        entry = {}
        for fmt in ("fp8", "bf16"):
            qs = stage_operands(qn, fmt)
            errs, sel_errs, per_frame_sd = [], [], []
            for c in range(0, args.M, 125_000):
                lc = ln[c:c + 125_000]
                exact = qn @ lc.t()
                pre = qs @ stage_operands(lc, fmt).t()
                errs.append((pre - exact).flatten())
                del exact, pre
            e = torch.cat(errs)                                        # [frames * M]
            entry[fmt] = {"all_pairs": moments(e)}
            del errs
            # selected sample: per frame the 64 rows with the best stage score
            pre_all = torch.cat([qs @ stage_operands(ln[c:c + 125_000], fmt).t() for c in range(0, args.M, 125_000)], 1)
            ex_all = torch.cat([qn @ ln[c:c + 125_000].t() for c in range(0, args.M, 125_000)], 1)
            top = torch.topk(pre_all, 64, dim=1).indices
            es = torch.gather(pre_all - ex_all, 1, top)
            entry[fmt]["top64_by_stage_score"] = {"mean": float(es.mean()), "std": float(es.std()),
                                                  "per_frame_mean_over_global_std": float((es.mean(1) / e.std()).mean()),
                                                  "per_frame_std_over_global_std_minmax": [float((es.std(1) / e.std()).min()),
                                                                                           float((es.std(1) / e.std()).max())]}
            # the rows that matter for a certificate: a frame's true top-4 -- how negative does their stage error get?
            t4 = torch.topk(ex_all, 4, dim=1).indices
            e4 = torch.gather(pre_all - ex_all, 1, t4)
            entry[fmt]["true_top4"] = {"mean": float(e4.mean()), "std": float(e4.std()), "min_over_global_std": float(e4.min() / e.std())}
            per_frame = (pre_all - ex_all).std(1)
            entry[fmt]["per_frame_std_minmax"] = [float(per_frame.min()), float(per_frame.max())]
            entry[fmt]["certificate_emulation"] = emulate_certificate(pre_all, ex_all, fmt, args.M)
            if args.centered:
                # the same stage on CENTRED operands: q.r = (q - mu).(r - mu) + [mu.r - mu.mu] + q.mu with mu = mean unit row;
                # the bracket is an exact per-row scalar, q.mu an exact per-frame scalar, only the first term is quantised
                mu = ln.mean(0, keepdim=True)
                qc, lc_ = qn - mu, ln - mu
                t_row = (ln @ mu.t()).flatten() - float((mu * mu).sum())
                u_f = (qn @ mu.t())
                qcs = stage_operands(qc, fmt)
                pre_c = torch.cat([qcs @ stage_operands(lc_[c:c + 125_000], fmt).t() for c in range(0, args.M, 125_000)], 1)
                pre_c = pre_c + t_row.unsqueeze(0) + u_f
                ec = pre_c - ex_all
                entry[fmt]["centred"] = {"mu_norm": float(mu.norm()), "rms_centred_row_norm": float(lc_.norm(dim=1).pow(2).mean().sqrt()),
                                         "all_pairs": moments(ec.flatten()),
                                         "certificate_emulation": emulate_certificate(pre_c, ex_all, fmt, args.M)}
                del pre_c, ec, qc, lc_
            del pre_all, ex_all, e
            torch.cuda.empty_cache()
        report["cases"][name] = entry
        print(name, json.dumps(entry), flush=True)
        del toks, feat, ln
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["audit", "zstats"])
    ap.add_argument("--M", type=int, default=1_000_000)
    ap.add_argument("-k", type=int, default=4)
    ap.add_argument("--cases", default="all")
    ap.add_argument("--modes", default="fp8,bf16")
    ap.add_argument("--frames", type=int, default=None, help="query frames of the cases that are not full-batch")
    ap.add_argument("--quick", action="store_true", help="--frames also for the full-batch cases")
    ap.add_argument("--centered", action="store_true", help="zstats: also the stage on mean-centred operands")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.what == "audit":
        args.frames = args.frames or 450 * 48
        r = run_audit(args)
        sys.exit(1 if r["total_mismatches"] else 0)
    args.frames = args.frames or 128
    if args.cases == "all":
        args.cases = "randn,dense,mixture,spiky,lowrank"
    run_zstats(args)


if __name__ == "__main__":
    main()
