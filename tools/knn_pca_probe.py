"""VERDICT r5 item 5 -- dense libraries, measured before anything is built.  A dense (single-speaker-like) bank defeats the fp6 / fp8
candidate stage because thousands of rows sit inside the stage's score error of a frame's 4th neighbour.  Proposal: take the bank's
common component out of both operands and add it back exactly,
    q . r  =  sum_{i < r} (q . u_i)(r . u_i)  +  q_perp . r_perp,
u_i the top-r principal directions of the UNIT rows (r = 0: nothing removed; "mean": the mean unit row only, as profiles/r03_knn_zstats
did), the first term in fp32 (r FMAs per score in the fold), only q_perp . r_perp on the low-precision MFMA -- with both residuals
RE-NORMALISED to unit length before they are quantised (block scales re-centre the e2m3 / e4m3 range; the norms |q_perp|, |r_perp|
travel as fp32 scalars), so that the stage error scales with |q_perp| |r_perp|.
For r in 0, mean, 1, 2, 4, 8, 16, 32 this script emulates, from full score matrices of `--frames` bench frames against the 1 M-vector
CE-derived bank (bench.ce_derived_tokens), what the search's first certificate would say: pass fraction at 7 sigma (per-frame sigma from
the frame's own 64 best candidates, floored by the stage prior scaled with the residual norms), the stage error, and how many rows a
7-sigma threshold pass would have to collect.  Decision rule of the verdict: build it if >= 50 % of the frames certify at some r.
  python tools/knn_pca_probe.py [--M 1000000] [--frames 2048] [--out profiles/r06_knn_pca_probe.json]"""
import argparse
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import knn_audit as A  # noqa: E402  (adds the package paths)


def e2m3(x):
    """round to nearest even onto the OCP e2m3 grid (1 sign, 2 exponent, 3 mantissa bits, bias 1), saturating at 7.5"""
    a = x.abs().clamp(max=7.5)
    step = torch.where(a < 2.0, torch.full_like(a, 0.125), torch.where(a < 4.0, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    q = torch.round(a / step) * step                  # torch.round is half-to-even
    return torch.sign(x) * q.clamp(max=7.5)


def stage(x_unit, fmt):
    b = x_unit.bfloat16().float()
    if fmt == "fp6":
        return e2m3(b * 32.0) / 32.0                  # knn.hip F6_SCALE
    return (b * 256.0).to(torch.float8_e4m3fn).float() / 256.0


def certificate(pre, exact, prior, M, k=4, z=7.0, splits=5, depth=16):
    """knn_audit.emulate_certificate with a per-frame prior (the floor of the per-frame sigma scales with the residual norms)"""
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
    cand = torch.cat(cand, 1)
    ce, cp = torch.gather(exact, 1, cand), torch.gather(pre, 1, cand)
    top = torch.topk(cp, min(64, cp.shape[1]), dim=1).indices
    err = torch.gather(cp - ce, 1, top)
    sig = torch.maximum(err.pow(2).mean(1).sqrt(), prior)
    vk = torch.topk(ce, k, dim=1).values[:, -1]
    true_vk = torch.topk(exact, k, dim=1).values[:, -1]
    passed = (vk - c) > z * sig
    n7 = (pre >= (vk - z * sig).unsqueeze(1)).sum(1).float()
    qs = torch.tensor([0.5, 0.9, 0.99], device=pre.device)
    return {"pass_fraction": round(float(passed.float().mean()), 4), "false_certifications": int((passed & (vk < true_vk - 1e-6)).sum()),
            "margin_vk_minus_c_median": round(float((vk - c).median()), 5), "sigma_median": float("%.3e" % sig.median()),
            "rows_within_7_sigma_of_v4_p50_p90_p99": [float(x) for x in torch.quantile(n7, qs)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=1_000_000)
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.pipeline import Converter
    dev = torch.device("cuda")
    conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev)
    toks, feat = A.make_case("dense", args.M, 450 * 384, dev, conv, bench)
    N, _, t = feat.shape
    flat = feat.permute(0, 2, 1).reshape(N * t, A.DIM)
    sel = torch.randperm(N * t, device=dev, generator=A.gen(dev, 5))[:args.frames]
    qn = A.normalise_rows(flat[sel])
    ln = A.normalise_rows(toks.t().contiguous())                       # [M, 768] unit rows
    M = ln.shape[0]
    exact = torch.cat([qn @ ln[c:c + 125_000].t() for c in range(0, M, 125_000)], 1)      # [F, M]
    # principal directions of the unit rows (uncentred second moment: the "common component" is its top direction)
    cov = (ln.double().t() @ ln.double()) / M
    evals, evecs = torch.linalg.eigh(cov)
    order = torch.argsort(evals, descending=True)
    evals, U = evals[order], evecs[:, order].float()                   # columns = directions
    report = {"M": M, "frames": args.frames, "library": "bench.ce_derived_tokens (dense, CE-derived)", "frames_from": "the bench batch",
              "second_moment_top_eigenvalues": [round(float(x), 5) for x in evals[:40]], "variants": {}}
    v4 = torch.topk(exact, 5, dim=1).values
    report["exact_v4_median"] = round(float(v4[:, 3].median()), 5)
    report["exact_gap_v4_v5_median"] = float("%.3e" % (v4[:, 3] - v4[:, 4]).median())
    variants = [("r0", None), ("mean", "mean")] + [(f"r{r}", r) for r in (1, 2, 4, 8, 16, 32)]
    for name, r in variants:
        if r is None:
            B = torch.zeros(A.DIM, 0, device=dev)
        elif r == "mean":
            mu = ln.mean(0, keepdim=True)
            B = (mu / mu.norm()).t().contiguous()                       # one direction: the normalised mean row
        else:
            B = U[:, :r].contiguous()
        aq, ar = qn @ B, ln @ B                                         # coefficients (fp32, exact part)
        qp, rp = qn - aq @ B.t(), ln - ar @ B.t()
        sq, sr = qp.norm(dim=1, keepdim=True).clamp(min=1e-20), rp.norm(dim=1, keepdim=True).clamp(min=1e-20)
        entry = {"directions_removed": int(B.shape[1]), "residual_norm_rows_median": round(float(sr.median()), 4),
                 "residual_norm_frames_median": round(float(sq.median()), 4)}
        for fmt, prior0 in (("fp6", 2.0e-3), ("fp8", 1.5e-3)):
            qs = stage(qp / sq, fmt)
            pre = torch.cat([(qs @ stage(rp[c:c + 125_000] / sr[c:c + 125_000], fmt).t()) * sr[c:c + 125_000].t()
                             for c in range(0, M, 125_000)], 1) * sq
            if B.shape[1]:
                pre = pre + torch.cat([aq @ ar[c:c + 125_000].t() for c in range(0, M, 125_000)], 1)
            err = pre - exact
            # the stage prior (floor of the per-frame sigma) scales like the error does: with the residual norms
            prior = prior0 * sq.flatten() * float(sr.median())
            e = certificate(pre, exact, prior, M)
            e["stage_error_std_all_pairs"] = float("%.3e" % err.std())
            t4 = torch.topk(exact, 4, dim=1).indices
            e["stage_error_std_true_top4"] = float("%.3e" % torch.gather(err, 1, t4).std())
            entry[fmt] = e
            del pre, err
            torch.cuda.empty_cache()
        report["variants"][name] = entry
        print(name, json.dumps(entry), flush=True)
    best = max((v[f]["pass_fraction"], n, f) for n, v in report["variants"].items() for f in ("fp6", "fp8"))
    report["best"] = {"pass_fraction": best[0], "variant": best[1], "format": best[2]}
    report["decision"] = ("BUILD: at least half of the frames certify" if best[0] >= 0.5 else
                          "CLOSE: no variant certifies half of the frames -- the bf16 stage stays the first stage of dense banks")
    print(json.dumps({"best": report["best"], "decision": report["decision"]}))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
