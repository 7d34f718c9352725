"""Upper bound of what prefetching weights into the Infinity Cache / L2 could buy at B = 1: the backbone with all 16 blocks sharing
ONE block's weight tensors (38 MB: resident in the 256 MB Infinity Cache once touched) against the real model, whose 0.66 GB of
backbone weights stream from HBM every image.  Same launches, same shapes; only the addresses differ.  (The ViT likewise.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev).contiguous()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


with torch.no_grad():
    ctx, _ = model.image_tokens(img)
    t_vit = timed(lambda: model.image_tokens(img))
    t_bb = timed(lambda: model.backbone_tokens(ctx))
    w = model._w
    keep_blocks, keep_layers = list(w["blocks"]), list(w["vit_layers"]) if "vit_layers" in w else None
    w["blocks"] = [keep_blocks[0]] * len(keep_blocks)
    t_bb_warm = timed(lambda: model.backbone_tokens(ctx))
    w["blocks"] = keep_blocks
    line = "backbone: real weights %.3f ms, all blocks on block 0's tensors %.3f ms (-%.3f ms)" % (t_bb, t_bb_warm, t_bb - t_bb_warm)
    for key in ("vit", "vit_layers", "layers"):
        if key in w and isinstance(w[key], list) and len(w[key]) > 1 and isinstance(w[key][0], dict):
            keep = list(w[key])
            w[key] = [keep[0]] * len(keep)
            t_vit_warm = timed(lambda: model.image_tokens(img))
            w[key] = keep
            line += "; tokenizer: %.3f -> %.3f ms (%s)" % (t_vit, t_vit_warm, key)
            break
    print(line)
    print("keys of the weight dict:", [k for k in w if isinstance(w[k], list)])
