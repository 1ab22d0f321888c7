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
    # ---- the BUILDABLE form (round 6): no per-vector scales, no change of the scoring kernel's geometry.  Unit vectors in the ROTATED
    # basis (coordinates ordered by eigenvalue): alpha = the top 32 coordinates, rho = the next 672, the 64 coordinates of least energy are
    # DROPPED so that K stays 768:   rows   [alpha_hi x 8 | alpha_lo x 128 | alpha_hi x 8 | rho x MR]
    #                                frames [alpha_hi x 8 | alpha_hi x 8  | alpha_lo x 128 | rho x MQ]      (e2m3 codes, MR / MQ powers of two)
    # with the four power-of-two weights 1/64, 1/1024, 1/1024, 1/(MR MQ) applied by the MFMA's block scales: the accumulator is the cosine
    # minus alpha_lo . alpha_lo and minus the dropped tail.
    evs = evals.float()
    report["tail_energy_last_64_of_768"] = float(evs[-64:].sum() / evs.sum())
    yq, yr = qn @ U, ln @ U                                              # rotated unit vectors
    aq, ar = yq[:, :32], yr[:, :32]
    rq, rr = yq[:, 32:704], yr[:, 32:704]

    def digits(a):
        hi = e2m3(a.bfloat16().float() * 8.0) / 8.0
        lo = e2m3((a - hi) * 128.0) / 128.0
        return hi, lo

    def pow2_mult(x, cap=7.5):
        m = float(x.abs().max())
        return 2.0 ** int(torch.floor(torch.log2(torch.tensor(cap / max(m, 1e-30)))))
    aqh, aql = digits(aq)
    arh, arl = digits(ar)
    MR = pow2_mult(rr)
    for MQ in (32.0, 64.0, 128.0):
        rqq = e2m3(rq * MQ) / MQ
        clipped = ((rq * MQ).abs() > 7.5).any(1)
        pre = torch.cat([aqh @ arh[c:c + 125_000].t() + aqh @ arl[c:c + 125_000].t() + aql @ arh[c:c + 125_000].t() +
                         rqq @ (e2m3(rr[c:c + 125_000] * MR) / MR).t() for c in range(0, M, 125_000)], 1)
        err = pre - exact
        prior = torch.full((args.frames,), 5.0e-4, device=dev)
        e = certificate(pre, exact, prior, M)
        keep = ~clipped
        e2 = certificate(pre[keep], exact[keep], prior[keep], M) if int(keep.sum()) > 16 else None
        e.update(stage_error_std_all_pairs=float("%.3e" % err.std()), stage_error_mean=float("%.3e" % err.mean()),
                 stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices).std()),
                 rows_multiplier=MR, frames_multiplier=MQ, frames_with_a_clipped_rho_element=int(clipped.sum()),
                 alpha_x8_max_rows=float((ar.abs() * 8).max()), alpha_x8_max_frames=float((aq.abs() * 8).max()),
                 pass_fraction_of_unclipped_frames=None if e2 is None else e2["pass_fraction"], prior=5.0e-4)
        report["variants"][f"buildable_r32_k768_mq{int(MQ)}"] = {"fp6": e, "fp8": {"pass_fraction": -1.0}}
        print(f"buildable MQ={MQ:g}", json.dumps(e), flush=True)
        del pre, err
        torch.cuda.empty_cache()
    # ---- the same with PER-BLOCK multipliers (one power of two per 32-coordinate block of the rotated basis, the same for every vector
    # of a side: block scales as per-k-step constants of the kernel) -- the coordinates of a block share an eigenvalue scale
    def block_mults(x, q=None):
        xb = x.abs().view(x.shape[0], -1, 32)
        m = xb.amax(dim=(0, 2)) if q is None else torch.quantile(xb.amax(dim=2), q, dim=0)
        return 2.0 ** torch.floor(torch.log2(7.5 / m.clamp(min=1e-30)))
    for name, qtl in (("max", None), ("p999", 0.999)):
        mr = block_mults(rr)                                       # rows: exact maxima (known when the library is packed)
        mq = block_mults(rq, qtl)                                  # frames: maxima / a quantile of a sample (clipped frames fall to the next tier)
        mr_e, mq_e = mr.repeat_interleave(32).unsqueeze(0), mq.repeat_interleave(32).unsqueeze(0)
        clipped = ((rq * mq_e).abs() > 7.5).any(1)
        rqq = e2m3(rq * mq_e) / mq_e
        pre = torch.cat([aqh @ arh[c:c + 125_000].t() + aqh @ arl[c:c + 125_000].t() + aql @ arh[c:c + 125_000].t() +
                         rqq @ (e2m3(rr[c:c + 125_000] * mr_e) / mr_e).t() for c in range(0, M, 125_000)], 1)
        err = pre - exact
        prior = torch.full((args.frames,), 5.0e-4, device=dev)
        keep = ~clipped
        e = certificate(pre[keep], exact[keep], prior[keep], M)
        e.update(stage_error_std_all_pairs=float("%.3e" % err[keep].std()), stage_error_mean=float("%.3e" % err[keep].mean()),
                 stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices)[keep].std()),
                 log2_rows_multipliers=[int(x) for x in torch.log2(mr)], log2_frames_multipliers=[int(x) for x in torch.log2(mq)],
                 frames_with_a_clipped_rho_element=int(clipped.sum()), prior=5.0e-4,
                 pass_fraction_counting_clipped_frames_as_failed=round(e["pass_fraction"] * float(keep.float().mean()), 4))
        report["variants"][f"buildable_r32_k768_block_scales_{name}"] = {"fp6": e, "fp8": {"pass_fraction": -1.0}}
        print(f"buildable block scales ({name})", json.dumps(e), flush=True)
        del pre, err
        torch.cuda.empty_cache()
    # ---- ... and with the rho coordinates MIXED by a fixed random rotation (eigen-coordinates concentrate the residual's energy in a few
    # blocks, which averages the quantisation error over fewer terms: sigma 1.06e-3 above against 3.7e-4 for the ideal form; a rotation
    # inside the rho subspace is free -- it is part of the one 768 x 768 basis matrix -- and makes every coordinate alike, so ONE
    # multiplier per side serves all rho blocks).  ra alpha directions in two e2m3 digits (3 ra codes), 768 - 3 ra rho coordinates.
    gR = torch.Generator(device=dev).manual_seed(99)
    for ra, with_ll in ((32, False), (32, True), (64, True)):
        nrho = 768 - (4 if with_ll else 3) * ra
        Rm = torch.linalg.qr(torch.randn(nrho, nrho, device=dev, generator=gR))[0]
        aq, ar = yq[:, :ra], yr[:, :ra]
        rq, rr = yq[:, ra:ra + nrho] @ Rm, yr[:, ra:ra + nrho] @ Rm
        aqh, aql = digits(aq)
        arh, arl = digits(ar)
        dropped = float(evs[ra + nrho:].sum() / evs.sum())
        MR = 2.0 ** int(torch.floor(torch.log2(torch.tensor(7.5 / float(rr.abs().max())))))
        for MQ in (32.0, 64.0):
            clipped = ((rq * MQ).abs() > 7.5).any(1)
            rqq = e2m3(rq * MQ) / MQ
            pre = torch.cat([aqh @ arh[c:c + 125_000].t() + aqh @ arl[c:c + 125_000].t() + aql @ arh[c:c + 125_000].t() +
                             (aql @ arl[c:c + 125_000].t() if with_ll else 0.0) +
                             rqq @ (e2m3(rr[c:c + 125_000] * MR) / MR).t() for c in range(0, M, 125_000)], 1)
            err = pre - exact
            prior = torch.full((args.frames,), 5.0e-4, device=dev)
            keep = ~clipped
            e = certificate(pre[keep], exact[keep], prior[keep], M)
            e.update(stage_error_std_all_pairs=float("%.3e" % err[keep].std()), stage_error_mean=float("%.3e" % err[keep].mean()),
                     stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices)[keep].std()),
                     rows_multiplier=MR, frames_multiplier=MQ, frames_with_a_clipped_rho_element=int(clipped.sum()),
                     energy_dropped=dropped, prior=5.0e-4, rho_coordinates=nrho,
                     pass_fraction_counting_clipped_frames_as_failed=round(e["pass_fraction"] * float(keep.float().mean()), 4))
            tag = f"mixed_a{ra}_{'4' if with_ll else '3'}blocks_mq{int(MQ)}"
            report["variants"][tag] = {"fp6": e, "fp8": {"pass_fraction": -1.0}}
            print("MIXED", tag, "pass", e["pass_fraction"], "counting clipped", e["pass_fraction_counting_clipped_frames_as_failed"], "sigma all", e["stage_error_std_all_pairs"],
                  "top4", e["stage_error_std_true_top4"], "MR", MR, "rows<=7sigma p50/p90/p99", e["rows_within_7_sigma_of_v4_p50_p90_p99"], "dropped", dropped, flush=True)
            del pre, err
            torch.cuda.empty_cache()
    # ---- ... and with ONE power-of-two scale per VECTOR for its rho part (the MX form: the row's scale byte travels in the padding of its
    # tile image and enters the MFMA as its block scale; the frame's sits in a register) -- no clipping by construction
    def vec_mult(x, cap=7.5):
        return 2.0 ** torch.floor(torch.log2(cap / x.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)))
    for ra, nblk in ((32, 3), (32, 4), (64, 4)):
        nrho = 768 - nblk * ra
        Rm = torch.linalg.qr(torch.randn(nrho, nrho, device=dev, generator=gR))[0]
        aq, ar = yq[:, :ra], yr[:, :ra]
        rq, rr = yq[:, ra:ra + nrho] @ Rm, yr[:, ra:ra + nrho] @ Rm
        aqh, aql = digits(aq)
        arh, arl = digits(ar)
        mq, mr = vec_mult(rq), vec_mult(rr)
        rqq = e2m3(rq * mq) / mq
        pre = torch.cat([aqh @ arh[c:c + 125_000].t() + aqh @ arl[c:c + 125_000].t() + aql @ arh[c:c + 125_000].t() +
                         (aql @ arl[c:c + 125_000].t() if nblk == 4 else 0.0) +
                         rqq @ (e2m3(rr[c:c + 125_000] * mr[c:c + 125_000]) / mr[c:c + 125_000]).t() for c in range(0, M, 125_000)], 1)
        err = pre - exact
        for pr in (5.0e-4, 3.0e-4):
            e = certificate(pre, exact, torch.full((args.frames,), pr, device=dev), M)
            e.update(stage_error_std_all_pairs=float("%.3e" % err.std()), stage_error_mean=float("%.3e" % err.mean()),
                     stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices).std()),
                     log2_rows_multiplier_min_median_max=[float(torch.log2(mr).min()), float(torch.log2(mr).median()), float(torch.log2(mr).max())],
                     log2_frames_multiplier_min_median_max=[float(torch.log2(mq).min()), float(torch.log2(mq).median()), float(torch.log2(mq).max())],
                     prior=pr, rho_coordinates=nrho, alpha_directions=ra, alpha_blocks=nblk)
            tag = f"mx_a{ra}_{nblk}blocks_prior{pr:g}"
            report["variants"][tag] = {"fp6": e, "fp8": {"pass_fraction": -1.0}}
            print("MX", tag, "pass", e["pass_fraction"], "sigma all", e["stage_error_std_all_pairs"], "top4", e["stage_error_std_true_top4"],
                  "rows<=7sigma", e["rows_within_7_sigma_of_v4_p50_p90_p99"], "log2 mr", e["log2_rows_multiplier_min_median_max"],
                  "log2 mq", e["log2_frames_multiplier_min_median_max"], flush=True)
        del pre, err
        torch.cuda.empty_cache()
    # ---- the form that needs NO kernel change: e4m3 (the fp8 stage) has range to spare, so every code is value x 256 as today and the four
    # alpha blocks are plain K elements -- rows [hi | lo | hi | lo | rho], frames [hi | hi | lo | lo | rho]: their dot product is
    # (hi + lo)(hi + lo) + rho . rho at the stage's usual 2^-16 weight.  Operand preparation only.
    def e4m3(x):
        return (x * 256.0).to(torch.float8_e4m3fn).float() / 256.0
    for ra, mix in ((32, True), (32, False), (64, True), (16, True)):
        nrho = 768 - 4 * ra
        aq, ar = yq[:, :ra], yr[:, :ra]
        rq, rr = yq[:, ra:ra + nrho], yr[:, ra:ra + nrho]
        if mix:
            Rm = torch.linalg.qr(torch.randn(nrho, nrho, device=dev, generator=gR))[0]
            rq, rr = rq @ Rm, rr @ Rm
        aqh = e4m3(aq); aql = e4m3(aq - aqh)
        arh = e4m3(ar); arl = e4m3(ar - arh)
        rqq = e4m3(rq)
        pre = torch.cat([(aqh + aql) @ (arh[c:c + 125_000] + arl[c:c + 125_000]).t() + rqq @ e4m3(rr[c:c + 125_000]).t() for c in range(0, M, 125_000)], 1)
        err = pre - exact
        for pr in (5.0e-4, 3.0e-4):
            e = certificate(pre, exact, torch.full((args.frames,), pr, device=dev), M)
            e.update(stage_error_std_all_pairs=float("%.3e" % err.std()), stage_error_mean=float("%.3e" % err.mean()),
                     stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices).std()),
                     prior=pr, rho_coordinates=nrho, alpha_directions=ra, rho_mixed=mix, energy_dropped=float(evs[ra + nrho:].sum() / evs.sum()))
            tag = f"fp8_digits_a{ra}_{'mixed' if mix else 'eigen'}_prior{pr:g}"
            report["variants"][tag] = {"fp8": e, "fp6": {"pass_fraction": -1.0}}
            print("F8", tag, "pass", e["pass_fraction"], "sigma all", e["stage_error_std_all_pairs"], "top4", e["stage_error_std_true_top4"],
                  "mean", e["stage_error_mean"], "rows<=7sigma", e["rows_within_7_sigma_of_v4_p50_p90_p99"], "dropped", e["energy_dropped"], flush=True)
        del pre, err
        torch.cuda.empty_cache()
    # ---- where the remaining error of the built form (fp8_digits_a64_mixed) sits: the leading coordinate alpha_0 (~0.65 for every row and
    # frame: its two-digit e4m3 value is good to ~2^-9 of 0.65) -- (i) alpha_0 exact, (ii) alpha_0 CENTRED: alpha_0 = c + a with c an
    # e4m3-exact constant near its mean, (c + a_q)(c + a_r) = c c + c a_r + a_q c + a_q a_r as nine code pairs instead of four
    ra, nrho = 64, 512
    Rm = torch.linalg.qr(torch.randn(nrho, nrho, device=dev, generator=torch.Generator(device=dev).manual_seed(7)))[0]
    aq, ar = yq[:, :ra], yr[:, :ra]
    rq, rr = yq[:, ra:ra + nrho] @ Rm, yr[:, ra:ra + nrho] @ Rm
    def two(x):
        h = e4m3(x)
        return h + e4m3(x - h)
    c0 = float(e4m3(ar[:, :1].mean().reshape(1, 1)))
    for name in ("a0_two_digits", "a0_exact", "a0_centred", "a0_a7_centred"):
        aq2, ar2 = two(aq), two(ar)
        if name == "a0_exact":
            aq2[:, 0], ar2[:, 0] = aq[:, 0], ar[:, 0]
        elif name in ("a0_centred", "a0_a7_centred"):
            ncen = 1 if name == "a0_centred" else 8
            for j in range(ncen):
                cj = float(e4m3(ar[:, j:j + 1].mean().reshape(1, 1)))
                aq2[:, j] = cj + two(aq[:, j:j + 1] - cj)[:, 0]
                ar2[:, j] = cj + two(ar[:, j:j + 1] - cj)[:, 0]
        rqq = e4m3(rq)
        pre = torch.cat([aq2 @ ar2[c:c + 125_000].t() + rqq @ e4m3(rr[c:c + 125_000]).t() for c in range(0, M, 125_000)], 1)
        err = pre - exact
        for pr in (3.5e-4, 2.5e-4):
            e = certificate(pre, exact, torch.full((args.frames,), pr, device=dev), M)
            e.update(stage_error_std_all_pairs=float("%.3e" % err.std()), prior=pr,
                     stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices).std()))
            report["variants"][f"fp8_digits_a64_{name}_prior{pr:g}"] = {"fp8": e, "fp6": {"pass_fraction": -1.0}}
            print("A0", name, "prior", pr, "pass", e["pass_fraction"], "sigma all", e["stage_error_std_all_pairs"], "top4", e["stage_error_std_true_top4"],
                  "rows<=7sigma", e["rows_within_7_sigma_of_v4_p50_p90_p99"], "margin", e["margin_vk_minus_c_median"], flush=True)
        del pre, err
        torch.cuda.empty_cache()
    # ---- the e2m3 (fp6, twice the MFMA rate) form of the built layout, leading coordinate centred: digits at block scales of their own
    # (hi x 8 | lo x 128), rho with one power-of-two multiplier per side -- what a block-scale table in the fp6 kernel would run
    def digits6(a):
        hi = e2m3(a * 8.0) / 8.0
        return hi, e2m3((a - hi) * 128.0) / 128.0
    c6 = float(e2m3(ar[:, :1].mean().reshape(1, 1) * 8.0) / 8.0)
    for MR6, MQ6 in ((64.0, 32.0), (128.0, 64.0), (64.0, 64.0)):
        aqc, arc = aq.clone(), ar.clone()
        aqc[:, 0] -= c6
        arc[:, 0] -= c6
        qh, ql = digits6(aqc)
        rh, rl = digits6(arc)
        clipped = ((rq * MQ6).abs() > 7.5).any(1)
        rows_clipped = int(((rr * MR6).abs() > 7.5).any(1).sum())
        rqq = e2m3(rq * MQ6) / MQ6
        a0q, a0r = (qh + ql)[:, :1], (rh + rl)[:, :1]
        pre = torch.cat([(qh + ql) @ (rh[c:c + 125_000] + rl[c:c + 125_000]).t() + c6 * c6 + c6 * a0q + c6 * a0r[c:c + 125_000].t() +
                         rqq @ (e2m3(rr[c:c + 125_000] * MR6) / MR6).t() for c in range(0, M, 125_000)], 1)
        err = pre - exact
        keep = ~clipped
        for pr in (5.0e-4, 3.5e-4):
            e = certificate(pre[keep], exact[keep], torch.full((int(keep.sum()),), pr, device=dev), M)
            e.update(stage_error_std_all_pairs=float("%.3e" % err[keep].std()), prior=pr, rows_multiplier=MR6, frames_multiplier=MQ6,
                     frames_with_a_clipped_rho_element=int(clipped.sum()), rows_with_a_clipped_rho_element=rows_clipped,
                     stage_error_std_true_top4=float("%.3e" % torch.gather(err, 1, torch.topk(exact, 4, dim=1).indices)[keep].std()))
            report["variants"][f"fp6_digits_a64_centred_mr{int(MR6)}_mq{int(MQ6)}_prior{pr:g}"] = {"fp6": e, "fp8": {"pass_fraction": -1.0}}
            print("F6C", MR6, MQ6, "prior", pr, "pass", e["pass_fraction"], "sigma all", e["stage_error_std_all_pairs"], "top4", e["stage_error_std_true_top4"],
                  "clipped frames", int(clipped.sum()), "clipped rows", rows_clipped, "rows<=7sigma", e["rows_within_7_sigma_of_v4_p50_p90_p99"], flush=True)
        del pre, err
        torch.cuda.empty_cache()
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
