"""Debug aid: run one step with the launch lists and with the persistent kernels, diff the two workspaces (fp32 words)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth
import mer_amd
from mer_amd.model import M2FNet

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_audio_only"
fwd_only = len(sys.argv) > 2 and sys.argv[2] == "fwd"
cfg, B, L, lengths, kind = synth.CASES[name]
sd = synth.make_state_dict(cfg)
batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn")]
ws = []
for mega in (0, 1):
    os.environ["M2F_MEGA"] = str(mega)
    torch.manual_seed(5)
    m = M2FNet(cfg, precision="bf16"); m.load_state_dict(sd); m = m.to("cuda:0").train()
    if fwd_only:
        eng = m.engine(); plan = eng.plan(B, L, True, False)
        plan.set_inputs(batch[0], batch[1], batch[2], batch[3]); plan.forward()
    else:
        m.train_step(*batch, use_graph=False)
    torch.cuda.synchronize()
    plan = next(iter(m.engine().plans.values()))
    print("mega", mega, "persistent", plan.persistent())
    try:
        plan.check_status()
    except Exception as e:
        print("STATUS", e)
    w = plan.workspace.clone()
    ws.append(w[plan._ws_off: plan._ws_off + (w.numel() - plan._ws_off) // 4 * 4].view(torch.int32))
    print("logits nan:", torch.isnan(plan.logits).sum().item(), "of", plan.logits.numel())
a, b = ws
n = min(a.numel(), b.numel())
d = (a[:n] != b[:n]).nonzero().flatten().cpu()
print("differing words:", d.numel(), "of", n)
if d.numel():
    # coalesce into ranges
    starts = [int(d[0])]; prev = int(d[0]); ranges = []
    for x in d.tolist()[1:]:
        if x > prev + 64:
            ranges.append((starts[-1], prev)); starts.append(x)
        prev = x
    ranges.append((starts[-1], prev))
    fa, fb = a[:n].view(torch.float32), b[:n].view(torch.float32)
    for s, e in ranges[:40]:
        seg = slice(s, e + 1)
        cnt = int((a[seg] != b[seg]).sum())
        print(f"  words [{s}, {e}] ({e - s + 1}): {cnt} differ; ref[{s}]={fa[s].item():.6g} new[{s}]={fb[s].item():.6g} nan_new={int(torch.isnan(fb[seg]).sum())}")
if name == "tiny_audio_only":
    T = B * L
    fa, fb = a.view(torch.float32), b.view(torch.float32)
    q_ref = fa[2112:2112 + T * 192].view(T, 192); q_new = fb[2112:2112 + T * 192].view(T, 192)
    x = batch[1].reshape(T, 64).to(torch.bfloat16).float()
    W = sd["audio_encoders.0.layers.0.self_attn.in_proj_weight"].cuda().to(torch.bfloat16).float()
    bias = sd["audio_encoders.0.layers.0.self_attn.in_proj_bias"].cuda()
    exp = x @ W.t() + bias
    print("ref vs expected:", (q_ref - exp).abs().max().item(), " new vs expected:", (q_new - exp).abs().max().item())
    print("new - bias vs x@W^T:", ((q_new - bias) - x @ W.t()).abs().max().item())
    print("new row0[:8]", q_new[0, :8].tolist()); print("exp row0[:8]", exp[0, :8].tolist())
    print("new col0[:8]", q_new[:8, 0].tolist()); print("exp col0[:8]", exp[:8, 0].tolist())
    # is new some permutation of exp?
    print("sorted diff:", (q_new.flatten().sort().values - exp.flatten().sort().values).abs().max().item())
    xt = exp.t().contiguous()
    ok = ((q_new - exp).abs() < 1e-5)
    for r in range(T):
        print(r, "".join("#" if ok[r, c * 8:(c + 1) * 8].all() else ("+" if ok[r, c * 8:(c + 1) * 8].any() else ".") for c in range(24)))
