"""Evaluation rule of the reference (src/train.py:260-272, src/test.py:61-74): accuracy and weighted-F1 are computed
PER BATCH with scikit-learn on the valid (label != -1) utterances, then averaged UNWEIGHTED over the batches."""
import torch
from sklearn.metrics import accuracy_score, f1_score


def move_batch(batch, device, non_blocking=False, text_encoder=None):
    """The four tensors of a collated batch on `device`: text, audio, emotion, padding_mask.
    text_encoder (a ``mer_amd.roberta.RobertaEncoder``; `runtime.text_encoder`, BASELINE config C5): the batch carries token ids
    ("text_ids" / "text_mask", [B, L, S]) and the text rows are computed HERE - the [CLS] hidden state of every valid utterance
    (src/feature_extractors/text/embeddings.py:83 of the reference, which dumps those rows to disk in a separate stage), under
    inference_mode (the reference fine-tunes the encoder in its own stage; the fusion model trains on its outputs), pads zero."""
    audio, emotion, mask = (batch[k].to(device, non_blocking=non_blocking) for k in ("audio", "emotion", "padding_mask"))
    if text_encoder is None or "text_ids" not in batch:
        return batch["text"].to(device, non_blocking=non_blocking), audio, emotion, mask
    ids, am = batch["text_ids"].to(device, non_blocking=non_blocking), batch["text_mask"].to(device, non_blocking=non_blocking)
    B, L, S = ids.shape
    valid = ~mask
    with torch.inference_mode():
        cls = text_encoder.cls_embeddings(ids[valid].contiguous(), am[valid].contiguous()).float()
    text = torch.zeros(B, L, cls.shape[1], dtype=torch.float32, device=device)
    text[valid] = cls.clone()
    return text, audio, emotion, mask


class BatchScores:
    def __init__(self):
        self.n_batches = 0
        self._acc = 0.0
        self._f1 = 0.0

    def update(self, logits: torch.Tensor, emotion: torch.Tensor) -> None:
        valid = emotion != -1
        predicted = logits.argmax(dim=2)[valid].reshape(-1).cpu().numpy()
        target = emotion[valid].reshape(-1).cpu().numpy()
        self._acc += accuracy_score(target, predicted)
        self._f1 += f1_score(target, predicted, average="weighted")
        self.n_batches += 1

    def sums(self):
        """(sum of per-batch accuracies, sum of per-batch weighted-F1s) - what ranks add up under data parallelism."""
        return self._acc, self._f1

    def result(self):
        """(accuracy, weighted_f1), each the plain mean of the per-batch scores."""
        n = max(self.n_batches, 1)
        return self._acc / n, self._f1 / n
