import cProfile, pstats, io, sys, time, torch
sys.path.insert(0, ".")
import sparch_amd
from sparch_amd import functional as Fn
dev = torch.device("cuda", 0)
B = 128
net = sparch_amd.SNN((B, None, 700), [512, 512, 20], neuron_type="adLIF", dropout=0.1, normalization="batchnorm").to(dev).train()
st = net.draw_states(B, dev)
for _ in range(20): net.draw_states_into(st, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): net.draw_states_into(st, B)
print("draw_states_into per call ms", 1e3 * (time.perf_counter() - t0) / 200)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): net.draw_states_into(st, B)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3000])
