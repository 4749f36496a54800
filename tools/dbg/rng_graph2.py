import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, bench, mer_amd
import test_mega_gpu as T
wl = bench.WORKLOADS["c2"]
cfg, B, L = dict(wl["cfg"], dropout=0.4), wl["B"], wl["L"]
sd = synth.make_state_dict(cfg)
batch = list(bench.synthetic_batch(cfg, B, L, 0, "cuda:0", ragged=True))
torch.manual_seed(11); ref = T._model(cfg, sd, mega=False)
torch.manual_seed(11); new = T._model(cfg, sd, mega=True)
for step in range(4):
    l0, z0, g0, p0 = T._step(ref, batch, use_graph=step > 0)
    r0 = ref.engine().rng.cpu().tolist()
    l1, z1, g1, p1 = T._step(new, batch, use_graph=step > 0)
    r1 = new.engine().rng.cpu().tolist()
    print(step, r0, r1, torch.equal(z0, z1), torch.equal(g0, g1), p0.persistent(), p1.persistent(), float(l0[0]), float(l1[0]))
