"""Run the attention fwd+bwd kernels many times on the bench shape (for rocprofv3 passes). usage: attn_one.py [B L H hd iters]"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import mer_amd
from mer_amd import functional as F
B, L, H, hd = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 16, 8, 96)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
E = H * hd
qkv = torch.randn(B * L, 3 * E, device="cuda")
q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
kp = torch.zeros(B, L, dtype=torch.bool, device="cuda")
dout = torch.randn(B * L, E, device="cuda")
for _ in range(iters):
    out, probs = F.attention_fwd(q, k, v, kp, B, L, H)
    F.attention_bwd(q, k, v, kp, out, probs, dout, B, L, H)
torch.cuda.synchronize()
