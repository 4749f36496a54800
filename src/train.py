"""Drop-in for the reference's ``src/train.py``: ``main``, ``training_loop``, ``train``, ``validate`` with the same
signatures, batch-dict keys, checkpoint format ({'epoch', 'model_state_dict', 'optimizer_state_dict'}), early
stopping / best-weights protocol and per-batch metric rule.  The model, criterion and optimizer are the HIP-backed
ones (M2FNet plan, fused CE, fused Adam); with ``runtime.fused_step`` the loop body of the reference
(src/train.py:227-231) runs as one hipGraph launch + one Adam kernel."""
import os
import sys
from datetime import datetime

import torch
from sklearn.metrics import accuracy_score, f1_score
from sklearn.utils import class_weight

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (_HERE, os.path.dirname(_HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from dataset import Dataset, collate_fn  # noqa: E402
from model import M2FNet  # noqa: E402
from utils import get_config  # noqa: E402
from mer_amd.optim import FusedAdam, M2FCrossEntropyLoss  # noqa: E402

try:
    from tqdm import tqdm
except ImportError:                                    # progress bars are cosmetic
    def tqdm(it, **_):
        return it
try:
    import wandb
except ImportError:
    wandb = None


def _runtime(cfg, key, default):
    rt = cfg.get("runtime", {}) if isinstance(cfg, dict) else getattr(cfg, "runtime", {})
    return rt.get(key, default) if rt else default


def main(config=None):
    config = get_config()
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    print(f"Using device {device}...")
    torch.manual_seed(int(_runtime(config, "seed", 0)))

    data_train = Dataset(mode="train")
    dl_train = torch.utils.data.DataLoader(data_train, collate_fn=collate_fn, **config.train.data_loader)
    data_val = Dataset(mode="val")
    dl_val = torch.utils.data.DataLoader(data_val, collate_fn=collate_fn, **config.val.data_loader)

    model = M2FNet(config.model, precision=_runtime(config, "precision", "fp32")).to(device)

    if config.solver.loss_fn != "CE":
        raise ValueError("Criterion not supported")
    weights = None
    if config.solver.balance_classes:
        w = class_weight.compute_class_weight(class_weight="balanced", classes=[0, 1, 2, 3, 4, 5, 6],
                                              y=data_train.get_labels())
        weights = torch.as_tensor(w, dtype=torch.float, device=device)
    criterion = M2FCrossEntropyLoss(weight=weights, ignore_index=-1, label_smoothing=0.1)

    optimizer = FusedAdam(model, lr=config.solver.lr, weight_decay=config.solver.weight_decay)

    if config.wandb.enabled:
        if wandb is None:
            raise RuntimeError("wandb.enabled is set but the wandb package is not installed")
        wandb.init(project=config.wandb.project_name, name=datetime.now().isoformat().split(".")[0], config=dict(config),
                   settings=wandb.Settings(start_method="spawn" if os.name == "nt" else "fork"),
                   entity=config.wandb.entity, resume="must" if config.wandb.resume_run else False,
                   id=config.wandb.resume_run_id)

    lr_scheduler = None
    if config.solver.scheduler.enabled:
        if config.solver.scheduler.scheduler_fn != "ExponentialLR":
            raise ValueError("Scheduler not supported")
        lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=config.solver.scheduler.gamma)

    start_epoch = 0
    load_path = os.path.abspath(config.checkpoint.load_path)
    if config.checkpoint.load_checkpoint and os.path.exists(load_path):
        checkpoint = torch.load(load_path, map_location=device)
        start_epoch = checkpoint["epoch"] + 1
        model.load_state_dict(checkpoint["model_state_dict"])
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])

    print("Training...")
    training_loop(model, dl_train, dl_val, criterion, optimizer, lr_scheduler, start_epoch, config, device)
    print("Training complete")


def _save(path, epoch, model, optimizer):
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict()}, path)


def training_loop(model, dl_train, dl_val, criterion, optimizer, lr_scheduler, start_epoch, config, device):
    solver = config.solver
    wandb_log = config.wandb.enabled
    save_path = os.path.abspath(config.checkpoint.save_path)
    os.makedirs(os.path.dirname(save_path), exist_ok=True)
    if wandb_log and config.wandb.watch_model:
        wandb.watch(model, criterion=criterion, log="all", log_freq=100, log_graph=False)

    stopping = solver.early_stopping.enabled
    best_path = os.path.join(os.path.dirname(save_path), "best_weights.pth")
    best_val, bad_epochs = float("inf"), 0
    train_losses, val_losses = [], []

    for epoch in range(start_epoch, solver.epochs):
        loss_train = train(model, dl_train, criterion, optimizer, epoch, wandb_log, device)
        train_losses.append(loss_train)
        loss_val, accuracy, weighted_f1 = validate(model, dl_val, criterion, device)
        val_losses.append(loss_val)

        if config.checkpoint.save_checkpoint:
            _save(save_path, epoch, model, optimizer)
        lr = optimizer.param_groups[0]["lr"]
        if solver.scheduler.enabled:
            lr_scheduler.step()
        print(f"Epoch: {epoch} lr: {lr:.3E} Train=[{loss_train:.3E}] Val=[{loss_val:.3E}] "
              f"Accuracy=[{accuracy * 100:.3f}%] Weighted_F1=[{weighted_f1 * 100:.3f}%]")
        if wandb_log:
            wandb.log({"Params/Epoch": epoch, "Params/Learning_Rate": lr, "Train/Loss": loss_train,
                       "Validation/Loss": loss_val, "Validation/Accuracy": accuracy,
                       "Validation/Weighted_F1": weighted_f1})

        if stopping:                                     # on validation LOSS, like the reference
            if loss_val < best_val:
                best_val, bad_epochs = loss_val, 0
                if solver.early_stopping.restore_best_weights:
                    _save(best_path, epoch, model, optimizer)
            else:
                bad_epochs += 1
                if bad_epochs >= solver.early_stopping.patience:
                    print(f"Early stopping: patience {solver.early_stopping.patience} reached")
                    if solver.early_stopping.restore_best_weights:
                        best = torch.load(best_path)
                        torch.save({k: best[k] for k in ("epoch", "model_state_dict", "optimizer_state_dict")}, save_path)
                        os.remove(best_path)
                        print(f"Best model at epoch {best['epoch']} restored")
                    break
    if wandb_log:
        wandb.finish()
    return {"loss_values": train_losses}


def _fusable(model, criterion):
    cfg = get_config() if os.path.exists("./src/config.yaml") else {}
    return (bool(_runtime(cfg, "fused_step", True)) and isinstance(criterion, M2FCrossEntropyLoss)
            and hasattr(model, "train_step"))


def train(model, dl_train, criterion, optimizer, epoch, wandb_log, device):
    loss_sum = 0.0
    model.train()
    fused = _fusable(model, criterion)
    use_graph = bool(_runtime(get_config(), "use_graph", True)) if fused and os.path.exists("./src/config.yaml") else True
    for idx_batch, batch in tqdm(enumerate(dl_train), total=len(dl_train), desc=f"Epoch {epoch}"):
        text = batch["text"].to(device, non_blocking=True)
        audio = batch["audio"].to(device, non_blocking=True)
        emotion = batch["emotion"].to(device, non_blocking=True)
        padding_mask = batch["padding_mask"].to(device, non_blocking=True)

        optimizer.zero_grad()
        if fused:
            loss = model.train_step(text, audio, padding_mask, emotion, label_smoothing=criterion.label_smoothing,
                                    class_weights=criterion.weight, use_graph=use_graph)
        else:
            outputs = model(text, audio, padding_mask)
            loss = criterion(outputs.permute(0, 2, 1), emotion)
            loss.backward()
        optimizer.step()
        loss_sum += loss.item()

        if wandb_log:
            wandb.log({"Train/Running_loss": loss_sum / (idx_batch + 1),
                       "Params/Global_step": epoch * len(dl_train) + idx_batch})
    return loss_sum / len(dl_train)


def validate(model, dl_val, criterion, device):
    loss_sum = accuracy = weighted_f1 = 0.0
    model.eval()
    with torch.inference_mode():
        for batch in tqdm(dl_val, total=len(dl_val), desc="Validation"):
            text = batch["text"].to(device)
            audio = batch["audio"].to(device)
            emotion = batch["emotion"].to(device)
            padding_mask = batch["padding_mask"].to(device)
            outputs = model(text, audio, padding_mask)
            loss_sum += criterion(outputs.permute(0, 2, 1), emotion).item()
            # per-batch scores, then an UNWEIGHTED mean over batches (reference src/train.py:261-272)
            keep = emotion != -1
            pred = torch.argmax(outputs, dim=2)[keep].flatten().cpu().numpy()
            true = emotion[keep].flatten().cpu().numpy()
            accuracy += accuracy_score(true, pred)
            weighted_f1 += f1_score(true, pred, average="weighted")
    n = len(dl_val)
    return loss_sum / n, accuracy / n, weighted_f1 / n


if __name__ == "__main__":
    main()
