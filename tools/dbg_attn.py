import ctypes, os, sys, torch
R = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, R)
import mer_amd
from mer_amd import runtime, functional as F
torch.manual_seed(0)
for (B, L, H, hd, ld) in ((32, 16, 5, 60, 904), (32, 16, 8, 96, 2304), (4, 9, 8, 128, 3072)):
    d = H * hd
    T = B * L
    ws = (torch.randn(T, ld, device="cuda") * 0.5)
    sh = ws.to(torch.bfloat16).contiguous()
    check = runtime.check
    check(runtime.lib().m2f_set_shadow_map(ws.data_ptr(), sh.data_ptr(), ws.numel()), "set_shadow_map")
    kp = torch.zeros(B, L, dtype=torch.bool, device="cuda")
    kp[1, L // 2:] = True
    q, k, v = ws[:, :d], ws[:, d:2 * d], ws[:, 2 * d:3 * d]
    os.environ["M2F_ATTN_BF16_KERNEL"] = "0"
    ref, _ = F.attention_fwd(q, k, v, kp, B, L, H)
    wr = sh.float()
    qr, kr, vr = wr[:, :d], wr[:, d:2 * d], wr[:, 2 * d:3 * d]
    check(runtime.lib().m2f_set_shadow_map(None, None, 0), "set_shadow_map")
    for mask, (a, b, c) in ((2, (qr, k, v)), (4, (q, kr, v)), (8, (q, k, vr)), (14, (qr, kr, vr))):
        want, _ = F.attention_fwd(a.contiguous(), b.contiguous(), c.contiguous(), kp, B, L, H)
        check(runtime.lib().m2f_set_shadow_map(ws.data_ptr(), sh.data_ptr(), ws.numel()), "set_shadow_map")
        os.environ["M2F_ATTN_BF16_KERNEL"] = str(mask)
        got, _ = F.attention_fwd(q, k, v, kp, B, L, H)
        os.environ["M2F_ATTN_BF16_KERNEL"] = "0"
        check(runtime.lib().m2f_set_shadow_map(None, None, 0), "set_shadow_map")
        valid = ~kp.reshape(-1)
        print(f"hd {hd} ld {ld} mask {mask}: vs same-rounding fp32 run {float((got - want)[valid].abs().max()):.3e}, vs unrounded {float((got - ref)[valid].abs().max()):.3e}")
