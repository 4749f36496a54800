"""Deterministic RoBERTa-shaped configs, weights and token batches for the text-encoder parity tests (numpy Philox, so
fixtures hold outputs only).  Keys / shapes are those of transformers.RobertaModel(add_pooling_layer=False).state_dict()."""
import numpy as np
import torch


def cfg(hidden, layers, heads, inter, vocab, max_pos):
    return {"hidden_size": hidden, "num_hidden_layers": layers, "num_attention_heads": heads, "intermediate_size": inter,
            "vocab_size": vocab, "max_position_embeddings": max_pos, "type_vocab_size": 1, "pad_token_id": 1,
            "layer_norm_eps": 1e-5, "hidden_act": "gelu", "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1}


# name -> (config, B, S, lengths)
CASES = {
    "roberta_tiny": (cfg(64, 2, 4, 128, 100, 40), 3, 19, [19, 7, 12]),
    "roberta_two_blocks": (cfg(128, 1, 2, 256, 120, 140), 2, 70, [70, 33]),          # hd = 64, keys span two 64-blocks
    "roberta_three_blocks": (cfg(96, 2, 3, 192, 90, 200), 2, 131, [131, 64]),       # hd = 32, three key blocks, ragged
    "roberta_base_width": (cfg(768, 1, 12, 3072, 300, 80), 2, 40, [40, 21]),          # roberta-base layer geometry
}


def state_dict_shapes(c):
    d, F = c["hidden_size"], c["intermediate_size"]
    sh = [("embeddings.word_embeddings.weight", (c["vocab_size"], d)),
          ("embeddings.position_embeddings.weight", (c["max_position_embeddings"], d)),
          ("embeddings.token_type_embeddings.weight", (c["type_vocab_size"], d)),
          ("embeddings.LayerNorm.weight", (d,)), ("embeddings.LayerNorm.bias", (d,))]
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            sh += [(p + f"attention.self.{n}.weight", (d, d)), (p + f"attention.self.{n}.bias", (d,))]
        sh += [(p + "attention.output.dense.weight", (d, d)), (p + "attention.output.dense.bias", (d,)),
               (p + "attention.output.LayerNorm.weight", (d,)), (p + "attention.output.LayerNorm.bias", (d,)),
               (p + "intermediate.dense.weight", (F, d)), (p + "intermediate.dense.bias", (F,)),
               (p + "output.dense.weight", (d, F)), (p + "output.dense.bias", (d,)),
               (p + "output.LayerNorm.weight", (d,)), (p + "output.LayerNorm.bias", (d,))]
    return sh


def make_state_dict(c, seed=7):
    sd = {}
    for idx, (name, shape) in enumerate(state_dict_shapes(c)):
        g = np.random.Generator(np.random.Philox(key=seed * 100003 + idx))
        x = g.standard_normal(shape).astype(np.float32)
        if name.endswith("LayerNorm.weight"):
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x = 0.05 * x
        elif "embeddings" in name:
            x = 0.5 * x
        else:
            x = x / np.sqrt(shape[1]) * 1.5          # keeps activations O(1) through the stack
        sd[name] = torch.from_numpy(np.asarray(x, dtype=np.float32))     # (the scaled matrices were float64 up to here)
    return sd


def make_batch(c, B, S, lengths, seed=11):
    g = np.random.Generator(np.random.Philox(key=seed))
    ids = g.integers(3, c["vocab_size"], size=(B, S)).astype(np.int64)
    ids[:, 0] = 0                                   # <s> = [CLS]
    mask = np.zeros((B, S), dtype=np.int64)
    for b, n in enumerate(lengths):
        mask[b, :n] = 1
        ids[b, n - 1] = 2                           # </s>
        ids[b, n:] = c["pad_token_id"]
    return torch.from_numpy(ids), torch.from_numpy(mask)
