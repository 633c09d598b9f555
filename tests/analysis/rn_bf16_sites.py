"""Analysis (test infrastructure, CPU only): which bf16 STORAGE POINT of the RawNet2 bf16 path costs the accuracy?

Runs the oracle's RawNet2 forward (oracle/rawnet2.py, reference RawNet2_custom.py:161-227) with a bf16 round-trip inserted at one
named site at a time (and at all of them), and prints cosine / max error of the embedding against the fp32 forward.  The sites are
the tensors the HIP bf16 handle stores as bf16: ln (LayerNorm output), filt (sinc filters), front, pre (lrelu(bn1)),
hb (conv1 output after bn2 + lrelu), o (conv2 + shortcut), x (gated block output), w (conv weights), att (attention hidden).

    python tests/analysis/rn_bf16_sites.py [n_utt] [seed_w]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import ecapa as o_e  # noqa: E402
from oracle import rawnet2 as o  # noqa: E402
from speakerverification_amd import synth  # noqa: E402


def r(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def forward(x, sd, sites, pool_shift=False):
    S = lambda n: n in sites or "all" in sites
    wq = lambda w: r(w, S("w"))
    x = r(o.layer_norm(x, sd), S("ln"))
    filt = r(o.sinc_filters(sd["first_conv.low_hz_"], sd["first_conv.band_hz_"]), S("filt"))
    x = F.conv1d(x.unsqueeze(1), filt.unsqueeze(1))
    x = F.max_pool1d(torch.abs(x), 3)
    x = r(F.leaky_relu(o.bn(x, sd, "first_bn"), 0.3), S("front"))
    for li, nblk in enumerate(o.LAYERS, start=1):
        for b in range(nblk):
            p = f"layer{li}.{b}"
            down = b == nblk - 1
            pre = r(F.leaky_relu(o.bn(x, sd, p + ".bn1"), 0.3), S("pre"))
            sc = F.conv1d(pre, wq(sd[p + ".shortcut.0.weight"])) if (p + ".shortcut.0.weight") in sd else x
            h = F.conv1d(pre, wq(sd[p + ".conv1.weight"]), padding=1)
            h = r(F.leaky_relu(o.bn(h, sd, p + ".bn2"), 0.3), S("hb"))
            out = r(F.conv1d(h, wq(sd[p + ".conv2.weight"]), padding=1) + sc, S("o"))
            if down:
                out = F.max_pool1d(out, 3)
            x = r(o.afms(out, sd, p + ".afms"), S("x"))
    x = r(F.leaky_relu(o.bn(x, sd, "bn_before_agg"), 0.3), S("pre") or S("agg"))
    a = F.conv1d(x, wq(sd["attention.0.weight"]), sd["attention.0.bias"])
    a = r(o.bn(F.leaky_relu(a, 0.01), sd, "attention.2"), S("att"))
    a = F.conv1d(a, wq(sd["attention.3.weight"]), sd["attention.3.bias"])
    w = F.softmax(a, dim=-1)
    m = torch.sum(x * w, dim=-1)
    s = torch.sqrt((torch.sum((x ** 2) * w, dim=-1) - m ** 2).clamp(min=1e-5))
    return F.linear(torch.cat([m, s], dim=1), sd["fc.weight"], sd["fc.bias"]), (m, s)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed_w = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sd = o_e.to_torch_sd(synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=seed_w))
    x = torch.from_numpy(synth.synth_waveforms(n, 32000, seed=20220829))
    torch.set_num_threads(8)
    with torch.no_grad():
        ref, (m0, s0) = forward(x, sd, ())
        print("scale", float(ref.abs().max()), "m scale", float(m0.abs().max()), "s range", float(s0.min()), float(s0.max()),
              "m/s median", float((m0.abs() / s0).median()))
        for sites in (("ln",), ("filt",), ("front",), ("pre",), ("hb",), ("o",), ("x",), ("w",), ("att",), ("agg",),
                      ("ln", "filt"), ("pre", "hb", "o", "x"), ("all",)):
            out, (m, s) = forward(x, sd, sites)
            cos = F.cosine_similarity(out, ref, dim=1)
            rel = float((out - ref).abs().max() / ref.abs().max())
            em = float((m - m0).abs().max() / m0.abs().max())
            es = float((s - s0).abs().max() / s0.abs().max())
            print(f"{'+'.join(sites):16s} cos min {float(cos.min()):.6f} mean {float(cos.mean()):.6f}  max err {rel:.4f} of scale   (m {em:.4f}  s {es:.4f})")


if __name__ == "__main__":
    main()
