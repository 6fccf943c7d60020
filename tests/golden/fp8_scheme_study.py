"""What operand format can buy on BASELINE configs[2] (fp8 encoder linears): the reference's own model (imported through
ref_harness.py -- build container only, never on the GPU box) with the inputs and weights of the four encoder linears per layer
quantised to OCP e4m3 under different scaling schemes, fp32 accumulation, everything else fp32; error = post-processed depth against
the unquantised run on the fixture frame (structured 1080p frame, seed 0, seeded synthetic weights).

    python tests/golden/fp8_scheme_study.py [vits|vitb|vitl]     -> table on stdout (kept in profiles/r4_fp8_scheme_study.md)

  bf16     operands rounded to bf16 (the bf16 engine's class, for scale)
  tensor   what the HIP e4m3 engine does: one static scale per activation tensor (amax / 448), one scale per weight output channel
  rows     one dynamic scale per activation row (token)
  mx       MX block scaling as v_mfma_scale_f32_*_f8f6f4 consumes it: one E8M0 (power-of-two) scale per 32 consecutive K elements,
           both operands
then each linear type alone under `tensor` and `mx`; with `--frontier` (round 5) the SUBSETS a mixed bf16 / e4m3 engine could run
(MLP only = 60 % of the encoder FLOPs, MLP + QKV, all four) and e5m2 (2 mantissa bits, 5 exponent bits) for the GELU output that
feeds FC2 -- the error half of the error / throughput frontier (profiles/r5_fp8_frontier.md)."""
import sys, os, contextlib, numpy as np, torch
sys.path.insert(0, "/root/repo/tests/golden"); sys.path.insert(0, "/root/repo")
from ref_harness import load_reference
from desktop2stereo_amd import synth
args = [a for a in sys.argv[1:] if not a.startswith("--")]
frontier = "--frontier" in sys.argv
four_k = "--4k" in sys.argv          # BASELINE configs[2]'s frame: 3840 x 2160 (CPU branch ::3 decimation) instead of 1080p
model = args[0] if args else "vits"
D = load_reference(model, 518, seed=0, fp32=True)
m = D.model_wraper.model
torch.set_num_threads(8)
img = synth.structured_frame(2160, 3840, 0) if four_k else synth.structured_frame(1080, 1920, 0)
x = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
xr = D._resize_patch_aligned_t(x, 518, 14); xn = xr / 255.0
mu, sd = D._normalization_tensors_for(xn); xn = (xn - mu) / sd
F8 = torch.float8_e4m3fn
def q_tensor(t, amax):  # per-tensor scale
    s = amax / 448.0
    return (t / s).clamp(-448, 448).to(F8).float() * s
def q_rows(t):   # per-row (last dim) dynamic
    s = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / 448.0
    return (t / s).clamp(-448, 448).to(F8).float() * s
def q_mx(t, blk=32):   # E8M0 scale per 32-element block along the last dim
    sh = t.shape; K = sh[-1]
    tb = t.reshape(-1, K // blk, blk)
    amax = tb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0))
    s = torch.pow(2.0, e)
    return ((tb / s).clamp(-448, 448).to(F8).float() * s).reshape(sh)
def q_bf16(t): return t.bfloat16().float()
lin = []
for lyr in m.backbone.encoder.layer:
    a = lyr.attention
    lin += [("qkv", a.attention.query), ("qkv", a.attention.key), ("qkv", a.attention.value), ("proj", a.output.dense), ("fc1", lyr.mlp.fc1), ("fc2", lyr.mlp.fc2)]
F8E5 = torch.float8_e5m2
def q_tensor_e5m2(t, amax):
    s = amax / 57344.0
    return (t / s).clamp(-57344, 57344).to(F8E5).float() * s
def run(mode, only=None, e5m2_for=()):
    hooks = []; saved = []
    for kind, l in lin:
        if only and kind not in only: continue
        W = l.weight.data
        saved.append((l, W.clone()))
        if mode == "tensor" or mode == "rows":
            s = W.abs().amax(dim=1, keepdim=True) / 448.0          # per-output-channel weight scales (as the engine)
            l.weight.data = (W / s).clamp(-448, 448).to(F8).float() * s
        elif mode == "mx":
            l.weight.data = q_mx(W)
        elif mode == "bf16":
            l.weight.data = q_bf16(W)
        def pre(mod, inp, mode=mode, kind=kind):
            t = inp[0]
            if mode == "tensor" and kind in e5m2_for: return (q_tensor_e5m2(t, t.abs().max()),)
            if mode == "tensor": return (q_tensor(t, t.abs().max()),)
            if mode == "rows": return (q_rows(t),)
            if mode == "mx": return (q_mx(t),)
            if mode == "bf16": return (q_bf16(t),)
        hooks.append(l.register_forward_pre_hook(pre))
    with torch.no_grad():
        raw = D.model_wraper(xn)
        post = D.post_process_depth(raw.float()).numpy()
    for h in hooks: h.remove()
    for l, W in saved: l.weight.data = W
    return post
with torch.no_grad():
    ref = D.post_process_depth(D.model_wraper(xn).float()).numpy()
if frontier:
    rows = {}
    for name, only, e5 in (("all four e4m3", None, ()), ("MLP only (FC1 + FC2)", ("fc1", "fc2"), ()), ("MLP + QKV", ("qkv", "fc1", "fc2"), ()),
                           ("MLP + proj", ("proj", "fc1", "fc2"), ()), ("QKV + proj only", ("qkv", "proj"), ()),
                           ("all four, GELU output (FC2 input) in e5m2", None, ("fc2",)), ("MLP only, GELU output in e5m2", ("fc1", "fc2"), ("fc2",))):
        p = run("tensor", only, e5); d = np.abs(p - ref)
        print(f"{model}{' 4k' if four_k else ''} frontier | {name:45s} | mean {d.mean():.5f} | max {d.max():.4f}", flush=True)
        rows[name] = {"mean": float(d.mean()), "max": float(d.max())}
    if "--json" in sys.argv:            # committed as tests/golden/fp8_frontier_<model>[_4k].json: the bounds tests/test_gpu_configs.py derives its e4m3 gates from
        import json
        out = os.path.join("/root/repo/tests/golden", f"fp8_frontier_{model}{'_4k' if four_k else ''}.json")
        with open(out, "w") as f:
            json.dump({"model": model, "frame": "structured_frame(2160, 3840, 0)" if four_k else "structured_frame(1080, 1920, 0)",
                       "what": "the REFERENCE's own model (ref_harness.py, seeded synthetic weights) with inputs and weights of the listed encoder linears "
                               "quantised to OCP e4m3 (static per-tensor activation scale, per-output-channel weight scale -- the HIP engine's scheme), fp32 "
                               "accumulation, everything else fp32: |post-processed depth - unquantised run|, depth range 1",
                       "rows": rows, "torch": torch.__version__}, f, indent=1)
        print("wrote", out)
    sys.exit(0)
for mode in ("bf16", "tensor", "rows", "mx"):
    p = run(mode); d = np.abs(p - ref)
    print(f"{model} {mode:7s}: post-depth mean {d.mean():.5f} max {d.max():.4f}", flush=True)
for only in (("qkv",), ("proj",), ("fc1",), ("fc2",)):
    p = run("tensor", only); d = np.abs(p - ref)
    p2 = run("mx", only); d2 = np.abs(p2 - ref)
    print(f"{model} only {only[0]:5s}: per-tensor mean {d.mean():.5f} | mx mean {d2.mean():.5f}", flush=True)
