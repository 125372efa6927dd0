"""The callers of Kevin's 2C model with the reference's signatures (example_scripts/Multimodal_example_task2C.py):

* ``KevinMultimodalDataset`` (:208-304) -- the organizers' dataset plus the caption stream: dict keys ``id, text, text_mask,
  caption_text, caption_text_mask, image[, label]``.  The reference generates the captions with BLIP inside ``__init__``
  (``ImageCaptioning``, :195-206, out of this path's scope): here they are passed in (``captions``: one string per sample, e.g. the
  reference's own BLIP output) or produced by a caller-supplied ``caption_fn(list of image paths) -> list of str``.
* ``train / test / evaluate`` (:688-871): the same loops -- focal loss called as ``criterion(output, labels, alpha=0.25, gamma=2.0,
  reduction='mean')``, gradient norm + clip, ``scheduler.step()`` per batch, accuracy from ``sigmoid(output) > 0.5``; ``test``
  returns ``(loss, accuracy, macro_f1, optimal_threshold)`` with the threshold at the ROC point of maximum ``tpr - fpr``
  (sklearn, as in the reference); ``evaluate`` writes the two TSVs (``id label run_id`` and ``id label prob run_id``).  Pinned to the reference's own functions run
  from its source: tests/golden/ref_kevin_2c.npz (oracle/gen_ref_kevin.py), tests/test_reference_run_gpu.py.

Differences, all on the host side of the hot path: the image transform (``Resize((224, 224))`` + flip + ColorJitter + rotation +
ToTensor + Normalize, :222-235) runs on the device through ``DeviceImagePipeline(mode="stretch", augment=training)`` on the decoded
uint8 images of a batch (``kevin_collate`` keeps them as a list); running loss / accuracy are accumulated on the device and read
once per epoch; the loaders of the reference's mid-epoch test / validation passes (globals ``test_df`` / ``val_df``, :756-769) are keyword
arguments of ``train``.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from .data import DeviceImagePipeline, HashTokenizer, id2l, synthetic_image

__all__ = ["KevinMultimodalDataset", "kevin_collate", "train", "test", "evaluate"]


class KevinMultimodalDataset(Dataset):
    """Same constructor order as the reference: (ids, text_data, image_data, labels, is_test=False)."""

    def __init__(self, ids: Sequence, text_data: Sequence, image_data: Sequence, labels: Optional[Sequence], is_test: bool = False,
                 captions: Optional[Sequence[str]] = None, caption_fn: Optional[Callable] = None, tokenizer=None,
                 english_tokenizer=None, max_seq_len: int = 128, image_size: int = 224, image_root: str = "",
                 synthetic_images: bool = False, vocab_size: int = 64000, english_vocab_size: int = 30522):
        self.ids, self.text_data, self.image_data = list(ids), list(text_data), list(image_data)
        self.labels = None if labels is None else list(labels)
        self.is_test = is_test
        self.tokenizer = tokenizer if tokenizer is not None else HashTokenizer(vocab_size)
        self.english_tokenizer = english_tokenizer if english_tokenizer is not None else HashTokenizer(english_vocab_size)
        self.max_seq_len, self.image_size, self.image_root, self.synthetic_images = max_seq_len, image_size, image_root, synthetic_images
        if captions is None:
            if caption_fn is None:
                raise ValueError("KevinMultimodalDataset needs `captions` (one string per sample) or a `caption_fn`: the BLIP captioner "
                                 "of the reference (ImageCaptioning) is outside this package")
            captions = caption_fn([os.path.join(image_root, p) for p in self.image_data])
        self.precalculated_captions = list(captions)
        if len(self.precalculated_captions) != len(self.ids):
            raise ValueError(f"{len(self.precalculated_captions)} captions for {len(self.ids)} samples")

    def __len__(self):
        return len(self.ids)

    def _encode(self, tok, s):
        e = tok.encode_plus(s, add_special_tokens=True, max_length=self.max_seq_len, padding="max_length", truncation=True,
                            return_attention_mask=True, return_tensors="pt")
        return e["input_ids"].squeeze(0), e["attention_mask"].squeeze(0)

    def __getitem__(self, index):
        text, text_mask = self._encode(self.tokenizer, self.text_data[index])
        cap, cap_mask = self._encode(self.english_tokenizer, self.precalculated_captions[index])
        path = os.path.join(self.image_root, self.image_data[index])
        if os.path.exists(path):
            from PIL import Image
            image = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)          # decoded, original size: the device resizes
        elif self.synthetic_images:
            image = (synthetic_image(str(self.ids[index]), self.image_size).permute(1, 2, 0) * 40 + 120).clamp(0, 255).to(torch.uint8).numpy()
        else:
            raise FileNotFoundError(f"{path} (pass synthetic_images=True to run without the image archive)")
        fdata = {"id": self.ids[index], "text": text, "text_mask": text_mask, "caption_text": cap, "caption_text_mask": cap_mask,
                 "image": image}
        if not self.is_test:
            fdata["label"] = torch.tensor(self.labels[index], dtype=torch.long)
        return fdata


def kevin_collate(items):
    """default_collate for the tensor fields; the decoded images (different sizes) and the ids stay lists."""
    out = {}
    for k in items[0]:
        vals = [it[k] for it in items]
        out[k] = torch.stack(vals) if isinstance(vals[0], torch.Tensor) else vals
    return out


def _images(data, device, pipeline: DeviceImagePipeline):
    img = data["image"]
    if isinstance(img, torch.Tensor):          # already a normalised f32 batch
        return img.to(device, non_blocking=True)
    return pipeline(img)


def _forward(model, data, device, pipeline):
    image = _images(data, device, pipeline)
    text, mask = data["text"].to(device), data["text_mask"].to(device)
    cap, cap_mask = data["caption_text"].to(device), data["caption_text_mask"].to(device)
    return model(text, image, mask, cap, cap_mask)


best_macro_f1 = 0.0          # the reference's module global (Multimodal_example_task2C.py:74,766-769)


def train(model, train_loader, criterion, optimizer, scheduler, device, epoch, scaler=None, image_pipeline: Optional[DeviceImagePipeline] = None,
          max_grad_norm: Optional[float] = None, eval_fn: Optional[Callable] = None, log_every: int = 10, test_df=None, val_df=None,
          stay_in_eval_mode_after_check: bool = True, evaluate_kwargs: Optional[dict] = None,
          clip_scaled_gradients: Optional[bool] = None) -> Tuple[float, float]:
    """Multimodal_example_task2C.py:688-776.

    ``test_df`` / ``val_df`` (the reference reads them as globals, :756-759): when given, the reference's mid-epoch check runs every
    ``total_batches // 2`` batches and after the last one -- ``test()`` on both loaders, the two report lines, and ``evaluate(model,
    test_df, threshold, device)`` whenever the test macro-F1 beats ``kevin.best_macro_f1`` (:766-769).  The reference's ``test()``
    leaves the model in eval mode and its ``train()`` never switches back, so every batch after the first check of an epoch is
    trained with BatchNorm on its running statistics and dropout off; ``stay_in_eval_mode_after_check=True`` (default) reproduces that,
    ``False`` returns to train mode.  ``eval_fn(batch_idx)`` is a free-form alternative to the two loaders (always followed by
    ``model.train()``).

    ``scaler``: a ``memehip.GradScaler`` (or None).  The fp16 build scales the 16-bit gradient streams inside the kernels; the scaler
    object carries the dynamic scale (halved on a skipped step, doubled after ``growth_interval`` clean ones) the way
    ``torch.cuda.amp.GradScaler`` does for the reference's fp16 branch (:712-717).  With ``memehip.Adam(..., max_grad_norm=...)`` the
    clip happens inside the fused update (one global norm); for any other optimizer ``max_grad_norm`` (reference: 1.0 under fp16, 10.0
    otherwise) is applied with ``clip_grad_norm_``.

    ``clip_scaled_gradients`` (default: True when a scaler is passed, else False): the reference's DEFAULT branch (``USE_FP16 = True``,
    :60) clips the gradients as ``scaler.scale(loss).backward()`` left them -- ``clip_grad_norm_(model.parameters(), 1.0)`` at :715
    with no ``unscale_`` before it -- and only then lets ``scaler.step`` divide by the scale (:716).  With torch's scale of 65536 the
    clipped TRUE gradient has norm <= 1.5e-5: the step is a fraction of the learning rate.  True reproduces exactly that (pinned to
    the reference run: tests/golden/ref_kevin_2c_fp16.npz); False unscales first and clips the true gradients, the corrected
    semantics.  With ``memehip.Adam`` the flag is written to ``optimizer.clip_scaled_gradients``."""
    global best_macro_f1
    from .model import Adam
    model.train()
    pipe = image_pipeline or DeviceImagePipeline(mode="stretch", augment=True, device=device)
    fused = isinstance(optimizer, Adam)
    if clip_scaled_gradients is None:
        clip_scaled_gradients = scaler is not None and hasattr(scaler, "scale")
    if fused:
        optimizer.clip_scaled_gradients = bool(clip_scaled_gradients)
        if scaler is not None and hasattr(scaler, "step") and hasattr(optimizer, "_attach_scaler") and getattr(scaler, "is_enabled", lambda: True)():
            if hasattr(scaler, "_optimizers"):
                optimizer._attach_scaler(scaler)          # before the first grad_norm(): the logged norm divides the scale out
    loss_sum = torch.zeros((), device=device)
    correct = torch.zeros((), device=device)
    total_batches = len(train_loader)
    check_interval = max(total_batches // 2, 1)
    window = []
    n_seen = 0
    for batch_idx, data in enumerate(train_loader, 1):
        optimizer.zero_grad()
        labels = data["label"].to(device).float()
        output = _forward(model, data, device, pipe)
        loss = criterion(output, labels, alpha=0.25, gamma=2.0, reduction="mean")
        if scaler is not None and hasattr(scaler, "scale"):
            scaler.scale(loss).backward()                      # :712
        else:
            loss.backward()
        if fused:
            grad_norm = optimizer.grad_norm() if (log_every and batch_idx % log_every == 0) else None
        else:
            if scaler is not None and hasattr(scaler, "unscale_") and not clip_scaled_gradients:
                scaler.unscale_(optimizer)                      # corrected semantics: clip the TRUE gradients (the reference clips the scaled ones, :713-715)
            grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), float("inf"))
            if max_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
        if scaler is not None and hasattr(scaler, "step"):
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()
        scheduler.step()
        loss_sum += loss.detach() * labels.size(0)
        window.append(loss.detach())
        n_seen += labels.size(0)
        if output.dim() == 1:
            predicted = (torch.sigmoid(output.detach()) > 0.5).float()
        else:
            predicted = torch.max(output.detach(), 1)[1].float()
        correct += (predicted == labels).sum()
        if log_every and batch_idx % log_every == 0:
            avg = float(torch.stack(window).mean())
            window = []
            print(f"TRAIN | Epoch [{epoch}] | Batch [{batch_idx}/{total_batches}] | Loss: {avg:.4f} | LR: {scheduler.get_last_lr()[0]} | "
                  f"Grad Norm: {float(grad_norm):.4f} |")
        if batch_idx % check_interval == 0 or batch_idx == total_batches:
            if test_df is not None and val_df is not None:
                t_loss, t_accuracy, t_macro_f1, t_optimal_threshold = test(model, test_df, criterion, device, epoch, image_pipeline=image_pipeline)
                v_loss, v_accuracy, v_macro_f1, v_optimal_threshold = test(model, val_df, criterion, device, epoch, image_pipeline=image_pipeline)
                print(f" TEST | Epoch [{epoch}] | Batch [{batch_idx}/{total_batches}] | Test Loss: {t_loss:.4f} | Acc: {t_accuracy:.4f} | "
                      f"F1: {t_macro_f1:.4f} | thresh: {t_optimal_threshold}")
                print(f" VAL | Epoch [{epoch}] | Batch [{batch_idx}/{total_batches}] | Test Loss: {v_loss:.4f} | Acc: {v_accuracy:.4f} | "
                      f"F1: {v_macro_f1:.4f} | thresh: {v_optimal_threshold}")
                if t_macro_f1 > best_macro_f1:
                    best_macro_f1 = t_macro_f1
                    evaluate(model, test_df, t_optimal_threshold, device, image_pipeline=image_pipeline, **(evaluate_kwargs or {}))
                if not stay_in_eval_mode_after_check:
                    model.train()
            if eval_fn is not None:
                eval_fn(batch_idx)
                model.train()
    n = len(train_loader.dataset)
    train_loss, accuracy = float(loss_sum) / n, float(correct) / n
    print(f"TRAIN | Epoch [{epoch}] | Training Loss: {train_loss:.4f} | Accuracy: {accuracy:.4f} |")
    return train_loss, accuracy


def test(model, test_loader, criterion, device, epoch, image_pipeline: Optional[DeviceImagePipeline] = None):
    """Multimodal_example_task2C.py:788-843 -> (test_loss, accuracy, macro_f1, optimal_threshold)."""
    from sklearn.metrics import f1_score, roc_curve
    model.eval()
    pipe = image_pipeline or DeviceImagePipeline(mode="stretch", augment=False, device=device)
    loss_sum = torch.zeros((), device=device)
    probs, truth = [], []
    with torch.no_grad():
        for data in test_loader:
            labels = data["label"].to(device).float()
            output = _forward(model, data, device, pipe)
            loss = criterion(output, labels, alpha=0.25, gamma=2.0, reduction="mean")
            loss_sum += loss * labels.size(0)
            probs.append(torch.sigmoid(output))
            truth.append(labels)
    predicted_probs = torch.cat(probs).float().cpu().numpy()
    true_labels = torch.cat(truth).cpu().numpy()
    fpr, tpr, thresholds = roc_curve(true_labels, predicted_probs)
    optimal_threshold = thresholds[np.argmax(tpr - fpr)]
    predicted = (predicted_probs > optimal_threshold).astype(float)
    n = len(test_loader.dataset)
    test_loss = float(loss_sum) / n
    accuracy = float((predicted == true_labels).sum()) / n
    macro_f1 = f1_score(true_labels, predicted, average="macro")
    print(f" TEST | Epoch [{epoch}] | Testing Loss: {test_loss:.4f} | Accuracy: {accuracy:.4f} | Macro F1: {macro_f1:.4f} | "
          f"optim t: {optimal_threshold} |")
    return test_loss, accuracy, macro_f1, optimal_threshold


def evaluate(model, test_loader, t_optimal_threshold, device, team_name: str = "memehip", run_id: Optional[str] = None, fold: int = 0,
             out_dir: str = ".", image_pipeline: Optional[DeviceImagePipeline] = None):
    """Multimodal_example_task2C.py:846-871: ``task2C_<team>.tsv`` (id, label, run_id) and ``task2C_<team>_probs_fold_<k>.tsv``
    (id, label, prob, run_id), labels from ``sigmoid(output) > t_optimal_threshold``.  Returns the two paths."""
    model.eval()
    pipe = image_pipeline or DeviceImagePipeline(mode="stretch", augment=False, device=device)
    predictions, probabilities, ids = [], [], []
    with torch.no_grad():
        for data in test_loader:
            prob = torch.sigmoid(_forward(model, data, device, pipe))
            predictions.append((prob > float(t_optimal_threshold)).float().cpu())
            probabilities.append(prob.float().cpu())
            ids.append(data["id"])
    run_id = run_id or f"{team_name}_memehip_2C.tsv"
    f1 = os.path.join(out_dir, f"task2C_{team_name}.tsv")
    f2 = os.path.join(out_dir, f"task2C_{team_name}_probs_fold_{fold}.tsv")
    with open(f1, "w") as f:
        f.write("id\tlabel\trun_id\n")
        for i, line in enumerate(predictions):
            for indx, l in enumerate(line.tolist()):
                f.write(f"{ids[i][indx]}\t{id2l[int(l)]}\t{run_id}\n")
    with open(f2, "w") as f:
        f.write("id\tlabel\tprob\trun_id\n")
        for i, line in enumerate(predictions):
            for indx, l in enumerate(line.tolist()):
                f.write(f"{ids[i][indx]}\t{id2l[int(l)]}\t{float(probabilities[i][indx])}\t{run_id}\n")
    return f1, f2
