import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights
from desktop2stereo_amd.vda_weights import make_vda_weights
dev = torch.device("cuda")
for model, H, W in [("vitb", 1080, 1920), ("vits", 720, 1280)]:
    cfg = MODELS[model]
    h, w, _ = engine_shape(H, W, 518)
    eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, 8, "bf16")
    p = PipelineParams(); sp = ops.sbs_params(0.064, 4.0, 0.0, "Full-SBS", False)
    frames = torch.from_numpy(np.stack([synth.structured_frame(H, W, i) for i in range(8)])).to(dev)
    ref = {}
    bad = 0
    for it in range(60):
        nb = [1, 3, 8, 2, 5][it % 5]
        out, d = eng.pipeline(frames[:nb], p, sp, want_depth=True)
        key = nb
        if key not in ref: ref[key] = (out.clone(), d.clone())
        else:
            if not torch.equal(out, ref[key][0]) or not torch.equal(d, ref[key][1]): bad += 1
    # per-frame result must not depend on the batch it rode in (beyond bf16 GEMM tile-order effects: compare depth loosely)
    d1 = ref[1][1][0]; d8 = ref[8][1][0]
    print(model, "nondeterministic repeats:", bad, "| batch-1 vs batch-8 depth max diff", float((d1 - d8).abs().max()))
    eng.close()
cfg = MODELS["vits"]
h, w, _ = engine_shape(1080, 1920, 336)
outs = []
for rep in range(2):
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), h, w, 1, "bf16", temporal=True)
    p = PipelineParams(depth_resolution=336); sp = ops.sbs_params(0.064, 4.0, 0.0, "Half-SBS", True)
    acc = []
    for i in range(45):
        f = torch.from_numpy(synth.structured_frame(1080, 1920, i % 7)).to(dev).unsqueeze(0)
        acc.append(eng.pipeline(f, p, sp).clone())
    outs.append(torch.stack(acc)); eng.close()
print("vda stream reproducible across engines:", bool(torch.equal(outs[0], outs[1])))
