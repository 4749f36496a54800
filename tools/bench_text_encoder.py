"""Secondary benchmark (SURVEY 8-f4 / BASELINE C5): forward of the in-loop text encoder (RoBERTa geometry, random weights -
the pretrained ones cannot be fetched offline) for the utterances of one M2FNet batch.  Prints one JSON line.
usage: python tools/bench_text_encoder.py [--model base|large] [--utterances 512] [--seq 64] [--dtype bf16|fp32] [--steps 10]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import mer_amd  # noqa: E402,F401
from mer_amd.roberta import RobertaEncoder  # noqa: E402

GEOM = {"base": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072),
        "large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large", choices=sorted(GEOM))
    ap.add_argument("--utterances", type=int, default=512)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--with-fusion-step", action="store_true",
                    help="BASELINE C5 data flow: encoder forward -> [CLS] rows -> one M2FNet training step (fwd+CE+bwd+Adam) per batch")
    a = ap.parse_args()
    cfg = dict(GEOM[a.model], vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
               layer_norm_eps=1e-5, hidden_act="gelu")
    torch.manual_seed(0)
    enc = RobertaEncoder(cfg, precision=a.dtype).cuda().eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, cfg["vocab_size"], (a.utterances, a.seq), generator=g)
    ids[:, 0] = 0
    ids = ids.cuda()
    step = lambda: enc.cls_embeddings(ids)                                    # noqa: E731
    fusion = None
    if a.with_fusion_step:
        import bench as B
        from mer_amd.model import M2FNet
        from mer_amd.optim import FusedAdam
        L = 16
        assert a.utterances % L == 0
        nb = a.utterances // L
        mcfg = B.model_cfg(768, cfg["hidden_size"], 768, 8, 8, 8, 6, 5)            # C3 geometry with d_text = encoder width
        model = M2FNet(mcfg, precision="bf16" if a.dtype == "fp8" else a.dtype).cuda().train()     # fp8 exists for the encoder only
        opt = FusedAdam(model, lr=5e-5, weight_decay=0.01)
        _, audio, mask, emotion = B.synthetic_batch(mcfg, nb, L, 0, torch.device("cuda"))
        fusion = {"dialogues": nb, "max_utt": L}

        def step():
            text = enc.cls_embeddings(ids).view(nb, L, -1)
            opt.zero_grad()
            model.train_step(text, audio, mask, emotion, label_smoothing=0.1)
            opt.step()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / a.steps
    d, F, Lr, H = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"]
    T = a.utterances * a.seq
    gemm = 2.0 * T * (4 * d * d + 2 * d * F) * Lr
    attn = 4.0 * a.seq * a.seq * (d // H) * a.utterances * H * Lr
    peak = {"bf16": 2500.0, "fp32": 157.3, "fp8": 5000.0}[a.dtype]
    print(json.dumps({"metric": ("utterances/sec, text encoder forward + M2FNet training step (BASELINE C5 data flow, RoBERTa-%s geometry)"
                                 if a.with_fusion_step else
                                 "utterances/sec, in-loop text encoder forward (RoBERTa-%s geometry, random weights)") % a.model,
                      "value": a.utterances / sec, "unit": "utterances/s", "ms_per_forward": sec * 1e3, "dtype": a.dtype,
                      "config": {"utterances": a.utterances, "tokens_per_utterance": a.seq, **GEOM[a.model]},
                      "algorithmic_tflop": (gemm + attn) / 1e12, "achieved_tflops": (gemm + attn) / sec / 1e12,
                      "frac_of_mfma_peak": (gemm + attn) / sec / 1e12 / peak, "peak_tflops": peak, "data": "synthetic token ids",
                      "with_fusion_step": fusion}))


if __name__ == "__main__":
    main()
