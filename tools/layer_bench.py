"""Time LayerDiffusion's layer stage (one launch per trajectory) at batch 64 / 400 steps, next to the CPU oracle."""
import sys
import time

import torch

sys.path.insert(0, ".")
from calodiffusion_amd.configs import load_config  # noqa: E402
from calodiffusion_amd.layerdiffusion import LayerDiffusion  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

cfg = load_config("dataset2")
torch.manual_seed(1234)
m = LayerDiffusion(cfg, n_steps=400, loss_type="l2")
for B in (64, 256, 1024):
    E, start = torch.rand((B, 1)).cuda(), torch.randn((B, 46)).cuda()
    m.sample_layers(E, start=start, sample_offset=0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        m.sample_layers(E, start=start, sample_offset=0)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    print(f"B={B}: {ms:.2f} ms per 400-step trajectory batch = {ms / 400 * 1e3:.1f} us/step, {B / ms * 1e3:.0f} layer vectors/s", flush=True)
om = O.OracleLayerModel(cfg, {k: v.cpu() for k, v in m.layer_model.state_dict().items()})
E, start = torch.rand((64, 1)), torch.randn((64, 46))
t0 = time.time()
with torch.no_grad():
    om.ddim_sample(start, E, None, 400)
print(f"CPU oracle B=64, 400 steps: {(time.time() - t0) * 1e3:.0f} ms ({torch.get_num_threads()} threads)")
