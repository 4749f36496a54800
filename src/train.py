"""Drop-in for the reference's ``src/train.py``: ``main``, ``training_loop``, ``train``, ``validate`` with the same
signatures, batch-dict keys, checkpoint format ({'epoch', 'model_state_dict', 'optimizer_state_dict'}), early
stopping / best-weights protocol and per-batch metric rule.  The model, criterion and optimizer are the HIP-backed
ones (M2FNet plan, fused CE, fused Adam); with ``runtime.fused_step`` the loop body of the reference
(src/train.py:227-231) runs as one hipGraph launch + one Adam kernel."""
import os
import sys
from datetime import datetime

import torch
from sklearn.utils import class_weight

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (_HERE, os.path.dirname(_HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from dataset import Dataset, DeviceLoader, collate_fn  # noqa: E402
from metrics import BatchScores, move_batch  # noqa: E402
from model import M2FNet  # noqa: E402
from utils import get_config  # noqa: E402
from mer_amd import dp  # noqa: E402
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

CHECKPOINT_KEYS = ("epoch", "model_state_dict", "optimizer_state_dict")
N_CLASSES = 7


def _runtime(cfg, key, default):
    """Value of the additive ``runtime:`` block of config.yaml (absent in the reference's file -> default)."""
    block = cfg.get("runtime", {}) if isinstance(cfg, dict) else getattr(cfg, "runtime", {})
    return block.get(key, default) if block else default


# ---- construction of the solver pieces from config.solver -----------------------------------------------------------
def build_criterion(solver, train_set, device):
    """CE(ignore_index=-1, label_smoothing=0.1), optionally with sklearn's balanced class weights of the train labels."""
    if solver.loss_fn != "CE":
        raise ValueError("Criterion not supported")
    class_weights = None
    if solver.balance_classes:
        balanced = class_weight.compute_class_weight(class_weight="balanced", classes=list(range(N_CLASSES)),
                                                     y=train_set.get_labels())
        class_weights = torch.as_tensor(balanced, dtype=torch.float, device=device)
    return M2FCrossEntropyLoss(weight=class_weights, ignore_index=-1, label_smoothing=0.1)


def build_scheduler(solver, optimizer):
    if not solver.scheduler.enabled:
        return None
    if solver.scheduler.scheduler_fn != "ExponentialLR":
        raise ValueError("Scheduler not supported")
    return torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=solver.scheduler.gamma)


TEXT_ENCODER_GEOMETRY = {"base": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072),
                         "large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)}


def build_text_encoder(te_cfg, d_text, device):
    """`runtime.text_encoder`: {enabled, precision: fp32 | bf16 | fp8, geometry: base | large | a dict of RobertaConfig fields,
    checkpoint: path of a state_dict in transformers.RobertaModel(add_pooling_layer=False) key layout (the encoder of the reference's
    TextERC, src/feature_extractors/text/model.py:16) or null = random weights}."""
    from mer_amd.roberta import RobertaEncoder
    geo = te_cfg.get("geometry", "base")
    cfg = dict(TEXT_ENCODER_GEOMETRY[geo]) if isinstance(geo, str) else dict(geo)
    for k, v in dict(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, hidden_act="gelu").items():
        cfg.setdefault(k, v)
    if cfg["hidden_size"] != d_text:
        raise ValueError(f"runtime.text_encoder produces {cfg['hidden_size']}-wide rows but model.TEXT.embedding_size is {d_text}")
    enc = RobertaEncoder(cfg, precision=te_cfg.get("precision", "bf16"))
    ck = te_cfg.get("checkpoint", None)
    if ck:
        enc.load_state_dict(torch.load(os.path.abspath(ck), map_location="cpu"))
    return enc.to(device).eval()


def start_wandb(config):
    if wandb is None:
        raise RuntimeError("wandb.enabled is set but the wandb package is not installed")
    run_name = datetime.now().isoformat().split(".")[0]
    wandb.init(project=config.wandb.project_name, name=run_name, config=dict(config), entity=config.wandb.entity,
               settings=wandb.Settings(start_method="spawn" if os.name == "nt" else "fork"),
               resume="must" if config.wandb.resume_run else False, id=config.wandb.resume_run_id)


def resume_if_requested(config, model, optimizer, device):
    """-> first epoch to run (0, or the checkpoint's epoch + 1)."""
    path = os.path.abspath(config.checkpoint.load_path)
    if not (config.checkpoint.load_checkpoint and os.path.exists(path)):
        return 0
    state = torch.load(path, map_location=device)
    model.load_state_dict(state["model_state_dict"])
    optimizer.load_state_dict(state["optimizer_state_dict"])
    return state["epoch"] + 1


def write_checkpoint(path, epoch, model, optimizer):
    torch.save(dict(zip(CHECKPOINT_KEYS, (epoch, model.state_dict(), optimizer.state_dict()))), path)


def main(config=None):
    config = get_config()
    # One process per GPU when launched under torchrun (WORLD_SIZE > 1) or with runtime.data_parallel: True - dialogues of every
    # global batch are sharded over the ranks, gradients are summed over RCCL with the GLOBAL valid-utterance denominator
    # (mer_amd/dp.py).  The reference is single-process (src/train.py:20); a single rank behaves exactly like it.
    want_dp = _runtime(config, "data_parallel", "auto")
    if want_dp not in (True, "auto") and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # W independent trainings on one device, all writing the same checkpoint, is never what a launcher was asked for
        raise RuntimeError(f"runtime.data_parallel is {want_dp!r} but this process was launched as one of "
                           f"{os.environ['WORLD_SIZE']} ranks (WORLD_SIZE): set runtime.data_parallel to True / 'auto' or "
                           "launch a single process")
    rank, world, local = dp.init_distributed() if want_dp in (True, "auto") else (0, 1, 0)
    device = torch.device(f"cuda:{local}" if torch.cuda.is_available() else "cpu")
    if rank == 0:
        print(f"Using device {device}..." + (f" ({world} ranks)" if world > 1 else ""))
    torch.manual_seed(int(_runtime(config, "seed", 0)))       # identical replicas and identical shuffles on every rank

    te_on = bool((_runtime(config, "text_encoder", None) or {}).get("enabled", False))
    if te_on:
        # the tokenizer is the caller's: a LOCAL directory with the files of the reference's RobertaTokenizer (text/dataset.py:9,42 fetch
        # 'roberta-base' by name - there is no network here)
        tok_dir = _runtime(config, "text_encoder").get("tokenizer", None)
        if not tok_dir:
            raise RuntimeError("runtime.text_encoder.enabled needs runtime.text_encoder.tokenizer: a local directory with the tokenizer files")
        from transformers import AutoTokenizer
        from dataset import tokenised_contexts
        tok = AutoTokenizer.from_pretrained(os.path.abspath(tok_dir))
        max_tokens = int(_runtime(config, "text_encoder").get("max_tokens", 64))
        train_set, val_set = Dataset(mode="train"), Dataset(mode="val")
        for ds_ in (train_set, val_set):
            ds_.token_ids, ds_.token_mask = tokenised_contexts(ds_.text, tok, max_tokens)
    else:
        train_set, val_set = Dataset(mode="train"), Dataset(mode="val")
    if _runtime(config, "device_batcher", False):          # embedding tables in HBM, one gather kernel per batch
        dl_train = DeviceLoader(train_set, device=device, seed=_runtime(config, "seed", 0), **config.train.data_loader)
        dl_val = DeviceLoader(val_set, device=device, **config.val.data_loader)
    else:
        gen = torch.Generator().manual_seed(int(_runtime(config, "seed", 0)))
        dl_train = torch.utils.data.DataLoader(train_set, collate_fn=collate_fn, generator=gen, **config.train.data_loader)
        dl_val = torch.utils.data.DataLoader(val_set, collate_fn=collate_fn, **config.val.data_loader)
    if world > 1:
        dl_train = dp.ShardedLoader(dl_train, rank, world)

    model = M2FNet(config.model, precision=_runtime(config, "precision", "fp32")).to(device)
    # how train() runs the loop body: (fused m2f_step instead of forward / criterion / backward, as one hipGraph)
    model.step_mode = (bool(_runtime(config, "fused_step", True)), bool(_runtime(config, "use_graph", True)))
    model.fused_optimizer = bool(_runtime(config, "fused_optimizer", False))
    if bool(_runtime(config, "grad_bf16", False)) and world == 1:
        model.set_grad_bf16(True)
    te_cfg = _runtime(config, "text_encoder", None)
    if te_cfg and te_cfg.get("enabled", False):
        # BASELINE config C5: the text rows are computed in the loop from token ids (Dataset(token_ids=...)) instead of read from
        # embeddings/<text>/<mode>.pkl.  Built OUTSIDE the fusion model (object.__setattr__: not a sub-module - its weights are
        # neither in M2FNet's state_dict / checkpoint format nor in the optimizer, as in the reference, which trains it in its own stage).
        object.__setattr__(model, "text_encoder", build_text_encoder(te_cfg, config.model.TEXT.embedding_size, device))
    criterion = build_criterion(config.solver, train_set, device)
    optimizer = FusedAdam(model, lr=config.solver.lr, weight_decay=config.solver.weight_decay)
    if world > 1:
        model.dp_step = dp.DataParallelStep(model, optimizer, n_buckets=int(_runtime(config, "grad_buckets", 4)),
                                            exchange=_runtime(config, "grad_exchange", "fp32"),
                                            overlap=bool(_runtime(config, "grad_overlap", False)),
                                            algorithm=_runtime(config, "grad_algorithm", "all_reduce"))
    if config.wandb.enabled and rank == 0:
        start_wandb(config)
    lr_scheduler = build_scheduler(config.solver, optimizer)
    first_epoch = resume_if_requested(config, model, optimizer, device)

    if rank == 0:
        print("Training...")
    training_loop(model, dl_train, dl_val, criterion, optimizer, lr_scheduler, first_epoch, config, device)
    if rank == 0:
        print("Training complete")
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def _rank():
    return torch.distributed.get_rank() if torch.distributed.is_initialized() else 0


# ---- early stopping on the validation LOSS with the reference's best-weights file protocol --------------------------
class EarlyStopper:
    def __init__(self, settings, save_path):
        self.enabled = bool(settings.enabled)
        self.patience = settings.patience
        self.restore = bool(settings.restore_best_weights)
        self.save_path = save_path
        self.best_path = os.path.join(os.path.dirname(save_path), "best_weights.pth")
        self.best_loss = float("inf")
        self.epochs_without_improvement = 0

    def should_stop(self, val_loss, epoch, model, optimizer) -> bool:
        if not self.enabled:
            return False
        if val_loss < self.best_loss:
            self.best_loss, self.epochs_without_improvement = val_loss, 0
            if self.restore and _rank() == 0:
                write_checkpoint(self.best_path, epoch, model, optimizer)
            return False
        self.epochs_without_improvement += 1
        if self.epochs_without_improvement < self.patience:
            return False
        if _rank() != 0:
            return True
        print(f"Early stopping: patience {self.patience} reached")
        if self.restore:                                 # the final checkpoint becomes the best one; the side file goes away
            best = torch.load(self.best_path)
            torch.save({k: best[k] for k in CHECKPOINT_KEYS}, self.save_path)
            os.remove(self.best_path)
            print(f"Best model at epoch {best['epoch']} restored")
        return True


def training_loop(model, dl_train, dl_val, criterion, optimizer, lr_scheduler, start_epoch, config, device):
    solver = config.solver
    rank0 = _rank() == 0
    log_to_wandb = config.wandb.enabled and rank0
    save_path = os.path.abspath(config.checkpoint.save_path)
    os.makedirs(os.path.dirname(save_path), exist_ok=True)
    if log_to_wandb and config.wandb.watch_model:
        wandb.watch(model, criterion=criterion, log="all", log_freq=100, log_graph=False)

    stopper = EarlyStopper(solver.early_stopping, save_path)
    history = {"loss_values": [], "val_loss_values": []}
    for epoch in range(start_epoch, solver.epochs):
        train_loss = train(model, dl_train, criterion, optimizer, epoch, log_to_wandb, device)
        val_loss, accuracy, weighted_f1 = validate(model, dl_val, criterion, device)
        history["loss_values"].append(train_loss)
        history["val_loss_values"].append(val_loss)

        if config.checkpoint.save_checkpoint and rank0:
            write_checkpoint(save_path, epoch, model, optimizer)
        lr_now = optimizer.param_groups[0]["lr"]
        if lr_scheduler is not None and solver.scheduler.enabled:
            lr_scheduler.step()
        if rank0:
            print(f"Epoch: {epoch} lr: {lr_now:.3E} Train=[{train_loss:.3E}] Val=[{val_loss:.3E}] "
                  f"Accuracy=[{accuracy * 100:.3f}%] Weighted_F1=[{weighted_f1 * 100:.3f}%]")
        if log_to_wandb:
            wandb.log({"Params/Epoch": epoch, "Params/Learning_Rate": lr_now, "Train/Loss": train_loss,
                       "Validation/Loss": val_loss, "Validation/Accuracy": accuracy, "Validation/Weighted_F1": weighted_f1})
        if stopper.should_stop(val_loss, epoch, model, optimizer):
            break
    if log_to_wandb:
        wandb.finish()
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.barrier()                      # rank 0 may still be moving checkpoint files
    return history


def _step_mode(model, criterion):
    """(fused, use_graph): the whole loop body as one m2f_step launch when the model / criterion are the HIP-backed ones.
    The two switches are read from the `runtime:` block ONCE, in main(), and travel on the model (`model.step_mode`)."""
    fused, use_graph = getattr(model, "step_mode", (True, True))
    return fused and isinstance(criterion, M2FCrossEntropyLoss) and hasattr(model, "train_step"), use_graph


def train(model, dl_train, criterion, optimizer, epoch, wandb_log, device):
    """One epoch; returns the mean of the per-batch losses (reference src/train.py:217-243)."""
    model.train()
    fused, use_graph = _step_mode(model, criterion)
    dp_step = getattr(model, "dp_step", None)             # set by main() when there is more than one rank
    running = 0.0
    progress = tqdm(enumerate(dl_train), total=len(dl_train), desc=f"Epoch {epoch}", disable=_rank() != 0)
    for step, batch in progress:
        text, audio, emotion, padding_mask = move_batch(batch, device, non_blocking=True, text_encoder=getattr(model, "text_encoder", None))
        if dp_step is not None:
            # sharded step: local sum-gradient -> RCCL all-reduce with the global denominator -> fused Adam, all inside
            loss = dp_step(text, audio, padding_mask, emotion, label_smoothing=criterion.label_smoothing,
                           class_weights=criterion.weight, use_graph=use_graph)
        else:
            optimizer.zero_grad()
            if fused and getattr(model, "fused_optimizer", False) and isinstance(optimizer, FusedAdam):
                # forward, criterion, backward AND optimizer.step() as one launch list (src/train.py:227-231 of the reference)
                loss = model.train_step(text, audio, padding_mask, emotion, label_smoothing=criterion.label_smoothing,
                                        class_weights=criterion.weight, use_graph=use_graph, optimizer=optimizer)
                running += loss.item()
                if wandb_log:
                    wandb.log({"Train/Running_loss": running / (step + 1), "Params/Global_step": epoch * len(dl_train) + step})
                continue
            if fused:
                loss = model.train_step(text, audio, padding_mask, emotion, label_smoothing=criterion.label_smoothing,
                                        class_weights=criterion.weight, use_graph=use_graph)
            else:
                loss = criterion(model(text, audio, padding_mask).permute(0, 2, 1), emotion)
                loss.backward()
            optimizer.step()
        running += loss.item()
        if wandb_log:
            wandb.log({"Train/Running_loss": running / (step + 1), "Params/Global_step": epoch * len(dl_train) + step})
    return running / len(dl_train)


def _rank_share(dl_val, rank, world):
    """Batches r, r + W, ... of the validation loader.  A torch DataLoader is re-built over exactly those index batches, so a rank
    neither collates nor uploads the batches it would discard; that needs the batches to be the same on every rank, i.e. no
    shuffling (the reference validates with shuffle: False, src/config.yaml:66).  Other loaders (DeviceLoader) are iterated and
    the foreign batches skipped."""
    if world == 1:
        return dl_val
    if isinstance(dl_val, torch.utils.data.DataLoader) and dl_val.batch_sampler is not None:
        if not isinstance(dl_val.sampler, torch.utils.data.SequentialSampler):
            raise ValueError("validation over several ranks deals whole batches to the ranks: val.data_loader.shuffle must be False")
        mine = list(dl_val.batch_sampler)[rank::world]
        return torch.utils.data.DataLoader(dl_val.dataset, batch_sampler=mine, collate_fn=dl_val.collate_fn,
                                           num_workers=dl_val.num_workers, pin_memory=dl_val.pin_memory)
    return (b for i, b in enumerate(dl_val) if i % world == rank)


def validate(model, dl_val, criterion, device):
    """-> (mean batch loss, accuracy, weighted_f1); scores by the per-batch rule of ``metrics.BatchScores``.
    With several ranks the replicas are identical and the rule is a plain mean over the reference's batches
    (src/train.py:245-272), so rank r evaluates batches r, r + W, ... WHOLE and the per-batch sums are added over the ranks:
    the same three numbers as the single-process loop, on every rank (early stopping decides alike everywhere)."""
    rank = _rank()
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    model.eval()
    scores, loss_total = BatchScores(), 0.0
    with torch.inference_mode():
        for batch in tqdm(_rank_share(dl_val, rank, world), total=(len(dl_val) - rank + world - 1) // world, desc="Validation", disable=rank != 0):
            text, audio, emotion, padding_mask = move_batch(batch, device, text_encoder=getattr(model, "text_encoder", None))
            logits = model(text, audio, padding_mask)
            loss_total += criterion(logits.permute(0, 2, 1), emotion).item()
            scores.update(logits, emotion)
    acc_sum, f1_sum = scores.sums()
    loss_total, acc_sum, f1_sum, n = dp.sum_over_ranks([loss_total, acc_sum, f1_sum, float(scores.n_batches)], device=device)
    n = max(n, 1.0)
    return loss_total / n, acc_sum / n, f1_sum / n


if __name__ == "__main__":
    main()
