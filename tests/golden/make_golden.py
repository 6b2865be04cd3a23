#!/usr/bin/env python3
"""Generate tests/golden/*.npz (float64, seeded numpy.random.default_rng) from an implementation of the op vocabulary.

    python tests/golden/make_golden.py                                   # --impl oracle  ->  tests/golden/oracle_ops.npz
    python tests/golden/make_golden.py --impl tf [--reference /path/to/iSeg] [--out tests/golden/tf_ops.npz]

The reference ships no golden vectors for this path and TensorFlow is not installable here, so `oracle_ops.npz` pins
the ORACLE (any later edit of oracle/ must reproduce it) and gives the GPU tests fixed inputs with fixed expected outputs that do not
depend on the oracle code at test time.  The SAME script, pointed at TensorFlow / Keras (and, for the ops that live in the reference
itself, at a checkout of it) with `--impl tf`, regenerates every vector from the real thing on the same seeded inputs:
tests/golden/impl_tf.py is the adapter (written against the TF >= 2.10 / Keras APIs the reference uses; it cannot run in this image).
Dropping the resulting `tf_ops.npz` into tests/golden/ turns on tests/test_oracle_known_answers.py::test_oracle_matches_tf_vectors and
upgrades "parity unpinned".  Only vectors travel -- never reference sources.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OUT = os.path.dirname(os.path.abspath(__file__))


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def npy(v):
    """result -> ndarray; None (op not provided by the adapter) stays None"""
    return None if v is None else (v.numpy() if hasattr(v, "numpy") else np.asarray(v))


def load_impl(name, reference=None):
    """(O, OM): two namespaces with the function names of oracle/tf_ops.py and oracle/models.py"""
    if name == "oracle":
        from oracle import models as OM
        from oracle import tf_ops as O

        return O, OM
    if name == "tf":
        sys.path.insert(0, OUT)
        import impl_tf

        return impl_tf.ops(reference), impl_tf.models(reference)
    raise SystemExit(f"unknown --impl {name!r} (oracle | tf)")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--impl", default="oracle")
    ap.add_argument("--reference", default=None, help="checkout of the reference (needed by --impl tf for its own ops: DCNv3, Swin tables, ...)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    O, OM = load_impl(args.impl, args.reference)
    out_path = args.out or os.path.join(OUT, "oracle_ops.npz" if args.impl == "oracle" else f"{args.impl}_ops.npz")
    rng = np.random.default_rng(0)
    g = {}
    # conv SAME with stride / dilation, odd sizes
    x = rng.standard_normal((2, 9, 7, 8))
    k = rng.standard_normal((3, 3, 8, 16)) / 8
    b = rng.standard_normal(16) * 0.1
    g["conv_x"], g["conv_k"], g["conv_b"] = x, k, b
    g["conv_s1d2"] = O.conv2d(t(x), t(k), t(b), 1, 2, "same").numpy()
    g["conv_s2d1"] = O.conv2d(t(x), t(k), t(b), 2, 1, "same").numpy()
    # depthwise 7x7 + LN + exact GELU (ConvNeXt block front half)
    xd = rng.standard_normal((1, 10, 11, 16))
    kd = rng.standard_normal((7, 7, 16, 1)) / 7
    g["dw_x"], g["dw_k"] = xd, kd
    yd = O.depthwise_conv2d(t(xd), t(kd), None, 1, 1, "same")
    g["dw_y"] = yd.numpy()
    gam, bet = rng.uniform(0.5, 1.5, 16), rng.standard_normal(16) * 0.1
    g["ln_gamma"], g["ln_beta"] = gam, bet
    g["ln_y"] = O.layer_norm(yd, t(gam), t(bet), 1e-6).numpy()
    g["gelu_y"] = O.gelu(yd).numpy()
    # bilinear resize x32-like (half-pixel, TF lerp order) and nearest label resize
    xr = rng.standard_normal((1, 3, 4, 5))
    g["resize_x"] = xr
    g["resize_y"] = O.resize_bilinear(t(xr), (13, 9)).numpy()
    lab = rng.integers(0, 21, (1, 6, 5, 1))
    g["nearest_x"] = lab.astype(np.int32)
    g["nearest_y"] = O.resize_nearest(torch.from_numpy(lab), (4, 9)).numpy().astype(np.int32)
    # ignore-label CE (mean over ALL pixels), first-max argmax, confusion matrix
    logits = rng.standard_normal((2, 4, 5, 21)) * 3
    labels = rng.integers(0, 21, (2, 4, 5))
    labels[rng.random((2, 4, 5)) < 0.2] = 255
    g["ce_logits"], g["ce_labels"] = logits, labels.astype(np.int32)
    g["ce_px"] = O.softmax_ce_ignore(torch.from_numpy(labels), t(logits), 21, 255).numpy()
    pred = O.argmax_first(t(logits))
    g["argmax"] = pred.numpy().astype(np.int32)
    g["confusion"] = O.confusion_matrix(torch.from_numpy(labels).reshape(-1), pred.reshape(-1), 21, 255).numpy()
    # BN training statistics (biased variance) + moving update
    xb = rng.standard_normal((2, 4, 4, 8)) * 2 + 1
    y, mean, var = O.batch_norm_train(t(xb), t(rng.uniform(0.5, 1.5, 8)), t(rng.standard_normal(8)), 1e-3)
    g["bn_x"], g["bn_mean"], g["bn_var"] = xb, mean.numpy(), var.numpy()
    # group norm / rms norm
    xg = rng.standard_normal((2, 3, 3, 12))
    g["gn_x"] = xg
    g["gn_y"] = O.group_norm(t(xg), None, None, 3, 1e-3).numpy()
    g["rms_y"] = npy(O.rms_norm(t(xg), t(np.zeros(12)), 1e-6))
    # pooling SAME
    xp = rng.standard_normal((1, 7, 6, 4))
    g["pool_x"] = xp
    g["maxpool_3s2"] = O.max_pool_same(t(xp), 3, 2).numpy()
    g["avgpool_2s2"] = O.avg_pool_same(t(xp), 2, 2).numpy()
    # bicubic (Keys a=-0.5, half-pixel, renormalised border taps, 1/1024 fraction table)
    xc = rng.standard_normal((1, 4, 4, 8))
    g["bicubic_x"] = xc
    g["bicubic_y"] = O.resize_bicubic(t(xc), (6, 5)).numpy()
    # DCNv3 core with the reference's [y,x]/[x,y] quirk
    xq = rng.standard_normal((1, 6, 5, 8))
    off = rng.standard_normal((1, 6, 5, 2 * 9 * 2)) * 1.2
    msk = torch.softmax(t(rng.standard_normal((1, 6, 5, 2, 9))), -1).reshape(1, 6, 5, 18).numpy()
    g["dcn_x"], g["dcn_off"], g["dcn_mask"] = xq, off, msk
    g["dcn_y"] = npy(O.dcnv3_op(t(xq), t(off), t(msk), (3, 3), (1, 1), "SAME", (1, 1), 2, 4, 1.0))
    # Swin shift mask and relative-position index
    sm, ri = npy(OM.swin_attention_mask(19, 23, 7, 3)), npy(OM._rel_index(7))
    g["swin_mask_19x23"] = None if sm is None else sm.astype(np.float32)
    g["swin_rel_index"] = None if ri is None else ri.astype(np.int32)
    # schedules / tiling
    lrs = [O.warmup_poly_decay(s, 1e-2, 30000, end_lr=0.0, warmup_steps=1500, warmup_lr=0.0, power=1.0) for s in (0, 500, 1000, 1500, 2000, 29999)]
    g["poly_lr"] = None if lrs[0] is None else np.array(lrs)
    for name, (ln, cr) in (("sliding_640_512", (640, 512)), ("sliding_1024_512", (1024, 512))):
        idx = O.sliding_start_indexs(ln, cr)
        g[name] = None if idx is None else np.array(idx)
    # AdamW (Keras decoupled weight decay) three steps
    w0, gr = rng.standard_normal(6), rng.standard_normal((3, 6))
    w, m, v = t(w0), torch.zeros(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)
    traj = []
    for s in range(3):
        res = O.adamw_step(w, t(gr[s]), m, v, s + 1, 1e-2, 1.0, 0.05)
        if res is None:
            traj = None
            break
        w, m, v = res
        traj.append(w.numpy().copy())
    g["adamw_w0"], g["adamw_g"], g["adamw_traj"] = w0, gr, (None if traj is None else np.stack(traj))
    g = {k: v for k, v in g.items() if v is not None}      # an adapter may not provide every op (returns None)
    np.savez_compressed(out_path, **g)
    print("wrote", out_path, sum(np.asarray(v).nbytes for v in g.values()), "bytes raw,", len(g), "arrays")


if __name__ == "__main__":
    main()
