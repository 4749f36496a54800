"""Evaluation rule of the reference (src/train.py:260-272, src/test.py:61-74): accuracy and weighted-F1 are computed
PER BATCH with scikit-learn on the valid (label != -1) utterances, then averaged UNWEIGHTED over the batches."""
import torch
from sklearn.metrics import accuracy_score, f1_score


def move_batch(batch, device, non_blocking=False):
    """The four tensors of a collated batch on `device`: text, audio, emotion, padding_mask."""
    return tuple(batch[k].to(device, non_blocking=non_blocking) for k in ("text", "audio", "emotion", "padding_mask"))


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
