"""Drop-in for the reference's ``src/test.py``: load ``checkpoint.load_path`` and report accuracy / weighted-F1
on the test split (mean of per-batch sklearn scores, reference src/test.py:51-74)."""
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (_HERE, os.path.dirname(_HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from dataset import Dataset, DeviceLoader, collate_fn  # noqa: E402
from metrics import BatchScores, move_batch  # noqa: E402
from model import M2FNet  # noqa: E402
from utils import get_config  # noqa: E402

try:
    from tqdm import tqdm
except ImportError:
    def tqdm(it, **_):
        return it


def load_model_weights(model, checkpoint_path, device):
    """Checkpoint format of the training loop: {'epoch', 'model_state_dict', 'optimizer_state_dict'}."""
    checkpoint_path = os.path.abspath(checkpoint_path)
    if not os.path.exists(checkpoint_path):
        raise ValueError("Checkpoint not found")
    state = torch.load(checkpoint_path, map_location=device)
    model.load_state_dict(state["model_state_dict"])
    return state.get("epoch")


def test(model, dl_test, device):
    """-> (accuracy, weighted_f1) over the loader, per-batch scores averaged unweighted."""
    scores = BatchScores()
    model.eval()
    with torch.inference_mode():
        for batch in tqdm(dl_test, total=len(dl_test)):
            text, audio, emotion, padding_mask = move_batch(batch, device, text_encoder=getattr(model, "text_encoder", None))
            scores.update(model(text, audio, padding_mask), emotion)
    return scores.result()


def main(config=None):
    config = get_config()
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    print(f"Using device {device}...")
    runtime_cfg = config.get("runtime", {}) or {}
    if runtime_cfg.get("device_batcher", False):
        loader = DeviceLoader(Dataset(mode="test"), device=device, **config.test.data_loader)
    else:
        loader = torch.utils.data.DataLoader(Dataset(mode="test"), collate_fn=collate_fn, **config.test.data_loader)
    model = M2FNet(config.model, precision=runtime_cfg.get("precision", "fp32")).to(device)
    load_model_weights(model, config.checkpoint.load_path, device)
    print("Testing...")
    accuracy, weighted_f1 = test(model, loader, device)
    print(f"Accuracy=[{accuracy * 100:.3f}%] Weighted_F1=[{weighted_f1 * 100:.3f}%]")
    print("Testing complete")


if __name__ == "__main__":
    main()
