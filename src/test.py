"""Drop-in for the reference's ``src/test.py``: load ``checkpoint.load_path`` and report accuracy / weighted-F1
on the test split (mean of per-batch sklearn scores, reference src/test.py:51-74)."""
import os
import sys

import torch
from sklearn.metrics import accuracy_score, f1_score

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (_HERE, os.path.dirname(_HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from dataset import Dataset, collate_fn  # noqa: E402
from model import M2FNet  # noqa: E402
from utils import get_config  # noqa: E402

try:
    from tqdm import tqdm
except ImportError:
    def tqdm(it, **_):
        return it


def main(config=None):
    config = get_config()
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    print(f"Using device {device}...")
    data_test = Dataset(mode="test")
    dl_test = torch.utils.data.DataLoader(data_test, collate_fn=collate_fn, **config.test.data_loader)
    rt = config.get("runtime", {}) or {}
    model = M2FNet(config.model, precision=rt.get("precision", "fp32")).to(device)
    path = os.path.abspath(config.checkpoint.load_path)
    if not os.path.exists(path):
        raise ValueError("Checkpoint not found")
    model.load_state_dict(torch.load(path, map_location=device)["model_state_dict"])
    print("Testing...")
    accuracy, weighted_f1 = test(model, dl_test, device)
    print(f"Accuracy=[{accuracy * 100:.3f}%] Weighted_F1=[{weighted_f1 * 100:.3f}%]")
    print("Testing complete")


def test(model, dl_test, device):
    accuracy = weighted_f1 = 0.0
    model.eval()
    with torch.inference_mode():
        for data in tqdm(dl_test, total=len(dl_test)):
            text, audio = data["text"].to(device), data["audio"].to(device)
            emotion, padding_mask = data["emotion"].to(device), data["padding_mask"].to(device)
            outputs = model(text, audio, padding_mask)
            keep = emotion != -1
            pred = torch.argmax(outputs, dim=2)[keep].flatten().cpu().numpy()
            true = emotion[keep].flatten().cpu().numpy()
            accuracy += accuracy_score(true, pred)
            weighted_f1 += f1_score(true, pred, average="weighted")
    return accuracy / len(dl_test), weighted_f1 / len(dl_test)


if __name__ == "__main__":
    main()
