"""Probe: does sampling two half batches on two HIP streams concurrently beat one full batch? (same weights, 400-step DDIM)"""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calodiffusion_amd.calodiffusion import CaloDiffusion
from calodiffusion_amd.configs import load_config

cfg = load_config("dataset2")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
def make():
    torch.manual_seed(1234)
    return CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
def inputs(B):
    g = torch.Generator().manual_seed(7)
    return torch.rand((B, 1), generator=g).cuda(), torch.randn((B, cfg["SHAPE_PAD"][2] + 1), generator=g).cuda()

def run_single(B, reps=2):
    m = make(); E, L = inputs(B)
    m.sample(E, layers=L, num_steps=N)  # warm-up / capture
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): m.sample(E, layers=L, num_steps=N)
    torch.cuda.synchronize(); return reps * B / (time.time() - t0)

def run_dual(B, reps=2):
    ms = [make(), make()]; ins = [inputs(B // 2), inputs(B // 2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for m, (E, L), s in zip(ms, ins, streams):
        with torch.cuda.stream(s): m.sample(E, layers=L, num_steps=N)
    torch.cuda.synchronize(); t0 = time.time()
    import threading
    def work(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(reps): ms[i].sample(ins[i][0], layers=ins[i][1], num_steps=N)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); return reps * B / (time.time() - t0)

print("single B=64:", run_single(64))
print("dual 2x32  :", run_dual(64))
print("single B=128:", run_single(128))
print("dual 2x64  :", run_dual(128))
