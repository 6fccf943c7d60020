python - <<'PY'
import os, time, torch, numpy as np, sys
sys.path.insert(0, os.getcwd())
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights
cfg = MODELS["vitb"]; h, w, _ = engine_shape(1080, 1920, 518)
for B in (32, 16, 12):
    frames = torch.from_numpy(np.stack([synth.noise_frame(1080, 1920, i) for i in range(B)])).cuda()
    p = PipelineParams(depth_resolution=518); sp = ops.sbs_params(0.064, 4.0, 0.0, "Full-SBS", False)
    eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, B, "bf16")
    for rep in range(2):
        for v in ("1", "0"):
            os.environ["D2S_LNF_PP"] = v; ops.reload_env()
            for _ in range(3): out = eng.pipeline(frames, p, sp)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): out = eng.pipeline(frames, p, sp)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"B={B} LNF_PP={v}: {dt*1e3:.3f} ms/step  {B/dt:.0f} fps", flush=True)
    eng.close()
PY
