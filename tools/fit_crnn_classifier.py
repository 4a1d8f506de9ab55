"""tools/fit_crnn_classifier.py [n_pages] [n_classes] [out.npz]: conditions the classifier of the synthetic CRNN so that its decisions have trained-like margins
(VERDICT r04 item 1b).

A seeded random-init ``Linear(512 -> 7644)`` over random BiLSTM features has 7644 near-tied logits per frame: the oracle's own top-2 margin is a few
1e-3 of the logit scale on most frames, so ANY 16-bit arithmetic flips token ids (bf16 reproduced 55 % of the fixture's strings) -- which says nothing
about what a user of a trained recogniser would see.  This script keeps the seeded random conv stack and BiLSTMs (crnn_state_dict(seed=1), the bench's and
the end-to-end fixture's), runs the CPU oracle (oracle/crnn.py, fp32) over the text lines of synthetic pages up to the classifier's input, clusters those
512-d frame features (k-means, K classes; the largest cluster -- the padding / background frames -- becomes CTC blank), and trains the K used rows of the
classifier with cross-entropy on the cluster labels (Adam, CPU, about a minute).  The other 7644 - K rows are zero (logit 0, never the arg-max).

What it is and is not: a WORKLOAD DEVICE, like tools/fit_layout_head.py.  The labels are clusters of the net's own features, not characters; the point is
the MARGIN DISTRIBUTION of the arg-max (printed below, before / after), which is what decides whether a rounding of the arithmetic changes a token id.
Output: pdf_table_amd/data/crnn_synth_classifier.npz (rows, their class ids, the fit's statistics); crnn_state_dict(conditioned=True) overlays it."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import crnn as ocrnn                                   # noqa: E402
from pdf_table_amd.synth_pages import make_page                    # noqa: E402
from pdf_table_amd.synth_weights import crnn_state_dict            # noqa: E402

SEED = 1          # bench.py's / the e2e fixture's recogniser checkpoint


def line_quads(idx, golden):
    """text-line quads of page idx: the oracle chain's detected boxes where the committed fixture has the page, else the generator's rectangles
    grown like a DB box (unclip of a h-pixel-high line adds ~ 0.3 h on every side)"""
    from e2e_synth import E2E_PAGES
    if golden is not None and idx in E2E_PAGES:
        return golden[f"p{E2E_PAGES.index(idx)}_det_boxes"].astype(np.float64)
    l = make_page(idx, 1024)[1]["lines"].astype(np.float64)
    g = np.round(0.3 * (l[:, 3] - l[:, 1]))
    x0, y0, x1, y1 = l[:, 0] - g, l[:, 1] - g, l[:, 2] + g, l[:, 3] + g
    return np.stack([x0, y0, x1, y0, x1, y1, x0, y1], 1).clip(0, 1023)


def features(sd, pages, golden, cache):
    """classifier inputs (the second BiLSTM's embedding output) of every frame of every line: [n_lines, 160, 512] fp32"""
    os.makedirs(cache, exist_ok=True)
    out = []
    for idx in pages:
        fn = os.path.join(cache, f"crnn_feat_p{idx}.npy")
        if not os.path.exists(fn):
            page = make_page(idx, 1024)[0]
            xs = torch.cat([ocrnn.rec_preprocess(ocrnn.crop_image(page, ocrnn.order_point(q))) for q in line_quads(idx, golden)])
            with torch.no_grad():
                f = ocrnn.crnn_features_fp32(sd, xs)
                r = ocrnn.bilstm_native(sd, "rnn.1", ocrnn.bilstm_native(sd, "rnn.0", f))       # [160, n, 512]
            np.save(fn, r.permute(1, 0, 2).contiguous().numpy())
        out.append(np.load(fn))
    return np.concatenate(out)


def margins(logits):
    t2 = torch.topk(logits, 2, dim=-1).values
    return ((t2[..., 0] - t2[..., 1]) / logits.abs().max()).flatten().numpy()


def describe(name, m):
    q = np.quantile(m, [0.01, 0.05, 0.25, 0.5])
    print(f"{name}: top-2 margin / logit scale -- 1 % {q[0]:.2e}, 5 % {q[1]:.2e}, 25 % {q[2]:.2e}, median {q[3]:.2e}; frames below 2.7e-3 (f16's measured "
          f"drift x 2) {float((m < 2.7e-3).mean()):.4f}, below 1.3e-2 (bf16's x 2) {float((m < 1.3e-2).mean()):.4f}")
    return {"q01": q[0], "q05": q[1], "q25": q[2], "q50": q[3], "below_f16": float((m < 2.7e-3).mean()), "below_bf16": float((m < 1.3e-2).mean())}


def main():
    n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    out_fn = sys.argv[3] if len(sys.argv) > 3 else os.path.join(REPO, "pdf_table_amd", "data", "crnn_synth_classifier.npz")
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    from e2e_synth import E2E_PAGES
    gfn = os.path.join(REPO, "tests", "golden", "e2e_page.npz")
    golden = np.load(gfn) if os.path.exists(gfn) else None
    pages = sorted(set(range(n_pages)) | set(E2E_PAGES))
    sd = {k: torch.as_tensor(np.asarray(v)).float() for k, v in crnn_state_dict(seed=SEED).items()}
    t0 = time.time()
    F = torch.from_numpy(features(sd, pages, golden, "/tmp/pt_fit_cache"))                 # [n, 160, 512]
    n = F.shape[0]
    X = F.reshape(-1, 512)
    print(f"{n} lines of pages {pages}: {X.shape[0]} frames, {time.time() - t0:.0f} s of oracle")
    before = describe("seeded random classifier", margins(F @ sd["cls.weight"].t()))

    # k-means on the frames (Lloyd, k-means++ style seeding by farthest points of a sample)
    g = torch.Generator().manual_seed(0)
    samp = X[torch.randperm(X.shape[0], generator=g)[:60000]]
    C = samp[:1].clone()
    d = ((samp - C[0]) ** 2).sum(1)
    for _ in range(K - 1):
        C = torch.cat([C, samp[int(torch.argmax(d))][None]])
        d = torch.minimum(d, ((samp - C[-1]) ** 2).sum(1))
    for it in range(25):
        a = torch.cdist(samp, C).argmin(1)
        for k in range(K):
            m = a == k
            if m.any():
                C[k] = samp[m].mean(0)
    lab = torch.cdist(X, C).argmin(1)
    cnt = torch.bincount(lab, minlength=K)
    blank = int(cnt.argmax())
    order = [blank] + [k for k in range(K) if k != blank]                      # cluster -> row: row 0 = blank
    remap = torch.empty(K, dtype=torch.long)
    remap[torch.tensor(order)] = torch.arange(K)
    y = remap[lab]
    print(f"k-means: {K} clusters, blank cluster holds {float(cnt[blank]) / len(lab):.3f} of the frames, smallest cluster {int(cnt.min())} frames")

    # the K used rows, cross-entropy on the cluster labels.  Initialised as the prototype classifier (w_k = c_k, no bias available: the reference's
    # Linear(512, 7644, bias=False), crnn/modeling_crnn.py:87), then trained so that the boundaries move into the gaps between clusters
    W = torch.nn.Parameter(C[torch.tensor(order)].clone() * 0.05)
    opt = torch.optim.Adam([W], lr=3e-3)
    for it in range(400):
        idx = torch.randint(0, X.shape[0], (32768,), generator=g)
        loss = torch.nn.functional.cross_entropy(X[idx] @ W.t(), y[idx])
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 100 == 99:
            with torch.no_grad():
                acc = float(((X @ W.t()).argmax(1) == y).float().mean())
            print(f"  iteration {it + 1}: loss {float(loss):.4f}, frames on their cluster's class {acc:.4f}")
    W = W.detach()
    # class ids: blank = 0 (CTC), the others spread over the vocabulary (one per 64-class tile of the classifier GEMM and beyond)
    ids = np.concatenate([[0], 1 + (np.arange(K - 1) * 79) % 7643]).astype(np.int32)
    assert len(set(ids.tolist())) == K
    full = torch.zeros(7644, 512)
    full[torch.from_numpy(ids).long()] = W
    lg = F @ full.t()
    after = describe("conditioned classifier", margins(lg))
    toks = lg.argmax(-1).numpy()
    coll = [int(((r[1:] != r[:-1]) & (r[1:] != 0)).sum() + (r[0] != 0)) for r in toks]
    print(f"decoded strings: {np.mean(coll):.1f} tokens per line (min {min(coll)}, max {max(coll)}); logit scale {float(lg.abs().max()):.1f}")
    np.savez_compressed(out_fn, rows=W.numpy().astype(np.float32), ids=ids, seed=np.array(SEED), pages=np.array(pages),
                        margin_before=np.array(list(before.values())), margin_after=np.array(list(after.values())),
                        margin_keys=np.array(list(after.keys())))
    print(out_fn, os.path.getsize(out_fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
