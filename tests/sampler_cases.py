"""The sampler cases of tests/golden/samplers_tiny.npz (made by oracle/gen_golden.py from the reference's own samplers), shared
by the oracle test (CPU) and the device test (GPU).  Each case: golden tag -> (sampler class name, config overrides,
SAMPLER_OPTIONS, sample_offset, batch rows)."""
import torch

CASES = {
    "euler_noisy": ("Euler", {"NOISY_SAMPLE": True}, None, 0, 3),
    "heun": ("Heun", {}, None, 0, 3),
    "heun_noisy": ("Heun", {"NOISY_SAMPLE": True}, None, 0, 3),
    "dpm2": ("DPM2", {}, None, 0, 3),
    "dpm2_off1": ("DPM2", {}, None, 1, 3),
    "lms": ("LMS", {}, None, 0, 3),
    "lms_o2": ("LMS", {}, {"ORDER": 2}, 0, 3),
    "restart_default": ("Restart", {}, None, 0, 3),
    "restart_int": ("Restart", {}, "restart_int", 0, 3),
    "restart_noisy": ("Restart", {"NOISY_SAMPLE": True}, "restart_noisy", 0, 3),
    # the reference's DPM family only broadcasts for batch 1 (gen_golden.py)
    "dpmpp2m": ("DPMPP2M", {}, None, 0, 1),
    "dpmpp2s": ("DPMPP2S", {}, None, 0, 1),
    "dpmpp2s_eta": ("DPMPP2S", {}, {"ETA": 1.0}, 0, 1),
    "dpm_7": ("DPM", {}, None, 0, 1),
    "dpm_6": ("DPM", {}, None, 0, 1),
    "dpm_2": ("DPM", {}, None, 0, 1),
    "consistency": ("Consistency", {"CONSIS_NSTEPS": 40}, None, 0, 3),
}


def options(g, tag):
    """SAMPLER_OPTIONS of a case (the Restart tables are stored in the golden)."""
    opts = CASES[tag][2]
    if opts == "restart_int":
        return {"RESTART_LIST": {int(k): [int(v[0]), int(v[1]), float(v[2]), float(v[3])]
                                 for k, v in zip(g["restart_int.keys"], g["restart_int.vals"])}}
    if opts == "restart_noisy":
        return {"RESTART_LIST": {3: [3, 1, 0.0, float(g["restart_noisy.tmax"])]}}
    return opts


def replay_noise(g, tag, shape):
    """The unit-normal tensors the reference sampler drew, in order: the CPU generator replayed from the recorded seed."""
    torch.manual_seed(int(g[f"{tag}.seed"]))
    return [torch.randn(shape) for _ in range(int(g[f"{tag}.draws"]))]
