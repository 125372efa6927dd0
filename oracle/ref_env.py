"""The environment the reference's own 2C code is RUN in to make fixtures -- and that the tests rebuild to feed the HIP path
the same inputs.  TEST INFRASTRUCTURE (see oracle/__init__.py): never imported by the product.

The reference scripts (example_scripts/Multimodal_example_task2C.txt / .py) cannot be imported whole: they need torchvision and
timm (not installed), download checkpoints and tokenizers, and train at module level.  ``extract()`` takes the definitions a
fixture needs out of the script's AST and executes exactly those, with the names they look up bound to what this container has:

=========================================  ==========================================================================================
name the reference uses                    bound to
=========================================  ==========================================================================================
``AutoModel.from_pretrained(name)``        a transformers ``DistilBertModel`` / ``BertModel`` built from an explicit local config (eager
                                           attention), weights from a seed (``seeded_*_state``)
``models.resnet50(pretrained=True)``       ``TorchvisionResNet50``: transformers' ``ResNetModel`` (bottleneck, stride on the 3x3 =
                                           torchvision v1.5) + ``fc = Linear(2048, 1000)`` -> the 1000 logits torchvision returns
``timm.create_model(name, pretrained)``    ``TimmViT``: transformers' ``ViTModel`` (LayerNorm eps 1e-6 as timm) with ``reset_classifier(0)``
                                           and token pooling -> the final-LayerNorm'd cls token
``AutoTokenizer.from_pretrained(name)``    ``EncodePlusTokenizer``: a real transformers ``BertTokenizer`` over a WordPiece vocabulary built
                                           from the texts at hand (``encode_plus`` of 4.39.2 forwards to ``__call__``; 5.x dropped the name)
``transforms`` (torchvision)               ``transforms`` below: Resize / CenterCrop / ToTensor / Normalize / RandomHorizontalFlip /
                                           ColorJitter / RandomRotation restated from torchvision 0.17.2's PIL code path
``sigmoid_focal_loss`` (torchvision.ops)   transformers' DETR helper (the other implementation of the detectron formula)
``tqdm``                                   identity
=========================================  ==========================================================================================

Inputs: 24 memes -- the 12 dev records of tests/golden/dev_sample12.json plus 12 made from them -- with seeded synthetic PNG images
of different sizes (the task's image archive is not in the repository).  Everything here is a pure function of seeds, so a test on
the GPU box rebuilds the same files / weights / batches without /root/reference.
"""
from __future__ import annotations

import ast
import json
import math
import os
import re
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ----------------------------------------------------------------------------------------------------------------------
# reference source -> namespace
# ----------------------------------------------------------------------------------------------------------------------
def extract(path: str, names: Iterable[str], namespace: dict) -> dict:
    """Execute the top-level class / function definitions ``names`` of the reference script ``path`` inside ``namespace``.
    Notebook shell lines (``!wget ...``) are blanked so that line numbers stay those of the file."""
    src = open(path, encoding="utf-8").read()
    src = "\n".join("" if ln.lstrip().startswith("!") else ln for ln in src.split("\n"))
    tree = ast.parse(src, filename=path)
    want = set(names)
    nodes = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    found = {n.name for n in nodes}
    assert found == want, f"missing in {path}: {sorted(want - found)}"
    namespace.setdefault("__name__", "reference_extract")
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), namespace)
    return namespace


# ----------------------------------------------------------------------------------------------------------------------
# data: 24 records + synthetic images
# ----------------------------------------------------------------------------------------------------------------------
def records24() -> List[dict]:
    """12 real dev records (texts, ids) + 12 derived ones; labels follow a fixed pattern with 9 of 24 positive (the dev sample
    itself has a single positive, too few for an ROC curve)."""
    base = json.load(open(os.path.join(GOLDEN, "dev_sample12.json"), encoding="utf-8"))
    pattern = [0, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0]
    out = []
    for k in range(24):
        a = base[k % 12]
        if k < 12:
            text = a["text"]
        else:                                   # second half of one text + first half of the next
            b = base[(k + 5) % 12]
            wa, wb = a["text"].split(), b["text"].split()
            text = " ".join(wa[len(wa) // 2:] + wb[: len(wb) // 2 + 1])
        out.append({"id": f"meme_{k:02d}", "img_path": f"img/meme_{k:02d}.png", "text": text,
                    "class_label": "propaganda" if pattern[k] else "not_propaganda"})
    return out


def synthetic_meme(k: int) -> np.ndarray:
    """uint8 [h, w, 3]: a coarse colour-block pattern + gradient, sizes between 180 and 420 pixels (both orientations)."""
    rng = np.random.default_rng(1000 + k)
    h = int(rng.integers(180, 421))
    w = int(rng.integers(180, 421))
    gh, gw = int(rng.integers(4, 10)), int(rng.integers(4, 10))
    coarse = rng.integers(0, 256, size=(gh, gw, 3)).astype(np.float32)
    yy = (np.arange(h) * gh // h)[:, None]
    xx = (np.arange(w) * gw // w)[None, :]
    img = coarse[yy, xx]
    ramp = (np.arange(h)[:, None] / h * 60.0 + np.arange(w)[None, :] / w * 40.0)[..., None]
    fine = rng.integers(-12, 13, size=(h, w, 3)).astype(np.float32)
    return np.clip(img * 0.7 + ramp + fine, 0, 255).astype(np.uint8)


def write_dataset(root: str) -> str:
    """Writes ``memes24.json`` and the PNGs under ``root``; returns the JSON path (img_path entries are relative to ``root``)."""
    from PIL import Image
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    recs = records24()
    for k, r in enumerate(recs):
        Image.fromarray(synthetic_meme(k)).save(os.path.join(root, r["img_path"]))
    path = os.path.join(root, "memes24.json")
    json.dump(recs, open(path, "w", encoding="utf-8"), ensure_ascii=False)
    return path


def captions24() -> List[str]:
    """What the (out-of-scope) BLIP captioner would return: one English sentence per meme, the reference's prompt first."""
    nouns = ["a man", "a cat", "two women", "a child", "a politician", "a crowd", "a dog", "a cartoon", "a soldier", "a teacher",
             "a football player", "an old man"]
    acts = ["looking at a phone", "laughing loudly", "holding a sign", "pointing at the camera", "sitting on a chair",
            "standing in the street", "talking to a friend", "reading a newspaper"]
    return [f"a meme of {nouns[k % 12]} {acts[(k * 5) % 8]} with arabic text" for k in range(24)]


# ----------------------------------------------------------------------------------------------------------------------
# tokenizer
# ----------------------------------------------------------------------------------------------------------------------
def wordpiece_vocab(texts: Sequence[str]) -> List[str]:
    """Specials + every character (whole and as a ## continuation) + every word of the texts (\\w+ runs; punctuation stands alone)."""
    words, chars = set(), set()
    for t in texts:
        for w in re.findall(r"\w+|[^\w\s]", t, flags=re.UNICODE):
            words.add(w)
            chars.update(w)
    chars = sorted(chars)
    return ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars + ["##" + c for c in chars] + sorted(w for w in words if len(w) > 1)


class EncodePlusTokenizer:
    """A transformers BertTokenizer over a local vocabulary behind the 4.39.2 method name the reference calls."""

    def __init__(self, texts: Sequence[str], workdir: str, tag: str = "vocab"):
        from transformers import BertTokenizer
        vocab = wordpiece_vocab(texts)
        path = os.path.join(workdir, f"{tag}.txt")
        open(path, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
        self.tok = BertTokenizer(path, do_lower_case=False)
        self.vocab_size = len(vocab)

    def encode_plus(self, text, **kw):
        return self.tok(text, **kw)

    def __call__(self, text, **kw):
        return self.tok(text, **kw)


# ----------------------------------------------------------------------------------------------------------------------
# torchvision.transforms (0.17.2, PIL code path) restated
# ----------------------------------------------------------------------------------------------------------------------
AUG_GEN = torch.Generator()          # the random transforms draw from this generator (torchvision: the global one)


class transforms:          # noqa: N801 -- stands where the module `torchvision.transforms` would
    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, img):
            for t in self.transforms:
                img = t(img)
            return img

    class Resize:
        """Resize(int): short side -> size, long side int(size * long / short); Resize((h, w)): exactly that.  PIL BILINEAR
        (antialiased, as torchvision does on PIL images)."""

        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            from PIL import Image
            w, h = img.size
            if isinstance(self.size, int):
                short, long = (w, h) if w <= h else (h, w)
                if short == self.size:
                    return img
                new_short, new_long = self.size, int(self.size * long / short)
                nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            else:
                nh, nw = self.size
            return img.resize((nw, nh), Image.BILINEAR)

    class CenterCrop:
        def __init__(self, size):
            self.size = (size, size) if isinstance(size, int) else tuple(size)

        def __call__(self, img):
            w, h = img.size
            th, tw = self.size
            assert th <= h and tw <= w, "padding branch of CenterCrop not restated (never taken: Resize(256) precedes it)"
            top = int(round((h - th) / 2.0))
            left = int(round((w - tw) / 2.0))
            return img.crop((left, top, left + tw, top + th))

    class ToTensor:
        def __call__(self, img):
            a = np.array(img, dtype=np.uint8)
            return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1), torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    class RandomHorizontalFlip:
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, img):
            from PIL import Image
            if float(torch.rand(1, generator=AUG_GEN)) < self.p:
                return img.transpose(Image.FLIP_LEFT_RIGHT)
            return img

    class ColorJitter:
        """get_params: a permutation of the four ops, then one uniform factor each from [1 - v, 1 + v] (hue: [-v, v]); the ops are
        torchvision's _functional_pil ones = PIL ImageEnhance (brightness, contrast, color) and the HSV hue shift."""

        def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
            self.b, self.c, self.s, self.h = brightness, contrast, saturation, hue
            self.last = None

        def draw(self):
            order = torch.randperm(4, generator=AUG_GEN).tolist()
            u = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi, generator=AUG_GEN))
            f = {0: u(max(0, 1 - self.b), 1 + self.b), 1: u(max(0, 1 - self.c), 1 + self.c), 2: u(max(0, 1 - self.s), 1 + self.s),
                 3: u(-self.h, self.h)}
            return order, f

        def __call__(self, img):
            order, f = self.draw()
            self.last = (order, f)
            return apply_color_jitter(img, order, f)

    class RandomRotation:
        """degrees=d -> angle ~ U(-d, d); F.rotate on a PIL image = img.rotate(angle, NEAREST, expand=False, fillcolor=0)."""

        def __init__(self, degrees):
            self.d = float(degrees)
            self.last = None

        def __call__(self, img):
            angle = float(torch.empty(1).uniform_(-self.d, self.d, generator=AUG_GEN))
            self.last = angle
            return rotate_pil(img, angle)


def adjust_hue_pil(img, hue_factor: float):
    """torchvision _functional_pil.adjust_hue: H channel of the HSV image shifted by uint8(hue_factor * 255) with wrap-around."""
    from PIL import Image
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.uint8(int(hue_factor * 255) & 0xFF)          # the C cast of np.uint8(float): truncate, wrap modulo 256
    h = Image.fromarray(np_h, "L")
    return Image.merge("HSV", (h, s, v)).convert("RGB")


def apply_color_jitter(img, order, f):
    from PIL import ImageEnhance
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(f[0])
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(f[1])
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(f[2])
        else:
            img = adjust_hue_pil(img, f[3])
    return img


def rotate_pil(img, angle: float):
    from PIL import Image
    return img.rotate(angle, Image.NEAREST, False, None, fillcolor=0)


def organizers_transform():
    T = transforms
    return T.Compose([T.Resize(256), T.CenterCrop(224), T.ToTensor(), T.Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))])


# ----------------------------------------------------------------------------------------------------------------------
# seeded weights (reference key names) and the third-party towers they are loaded into
# ----------------------------------------------------------------------------------------------------------------------
_B2D = (("attention.self.query.", "attention.q_lin."), ("attention.self.key.", "attention.k_lin."),
        ("attention.self.value.", "attention.v_lin."), ("attention.output.dense.", "attention.out_lin."),
        ("attention.output.LayerNorm.", "sa_layer_norm."), ("intermediate.dense.", "ffn.lin1."),
        ("output.dense.", "ffn.lin2."), ("output.LayerNorm.", "output_layer_norm."))


def bert_to_distil_name(k: str) -> str:
    """BertModel parameter name -> DistilBertModel parameter name (transformers 4.39.2 and the installed 5.x agree)."""
    if k.startswith("encoder.layer."):
        k = "transformer.layer." + k[len("encoder.layer."):]
        for a, b in _B2D:
            if a in k:
                return k.replace(a, b)
    return k


def seeded_text_state(vocab_size: int, layers: int, seed: int, hidden: int = 768, heads: int = 12, intermediate: int = 3072,
                      max_position: int = 512, type_vocab: int = 2, naming: str = "bert") -> Dict[str, torch.Tensor]:
    """BertModel / DistilBertModel state (no prefix) drawn by oracle.meme_oracle.init_params' rule (N(0, 0.02), LayerNorm
    gamma 1 + N(0, 0.02))."""
    from . import meme_oracle as O
    tc = O.TextConfig(vocab_size=vocab_size, hidden=hidden, layers=layers, heads=heads, intermediate=intermediate,
                      max_position=max_position, type_vocab=type_vocab)
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in O._text_shapes(tc, pfx="").items():
        t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
        if "LayerNorm.weight" in name:
            t = t + 1.0
        out[bert_to_distil_name(name) if naming == "distilbert" else name] = t
    return out


def seeded_vit_state(seed: int, image_size: int = 224, patch: int = 16, hidden: int = 768, layers: int = 12, heads: int = 12,
                     intermediate: int = 3072) -> Dict[str, torch.Tensor]:
    """ViT state with the oracle's (transformers 4.39.2 ViTModel) names, no prefix."""
    from . import meme_oracle as O
    ic = O.ImageConfig(image_size=image_size, patch=patch, hidden=hidden, layers=layers, heads=heads, intermediate=intermediate)
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in O._image_shapes(ic, pfx="").items():
        t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
        if "layernorm" in name and name.endswith("weight"):
            t = t + 1.0
        out[name] = t
    return out


def seeded_linear(out_f: int, in_f: int, g: torch.Generator):
    b = 1.0 / math.sqrt(in_f)
    return (torch.rand((out_f, in_f), generator=g) * 2 - 1) * b, (torch.rand((out_f,), generator=g) * 2 - 1) * b


def seeded_bn(n: int, g: torch.Generator, pfx: str) -> Dict[str, torch.Tensor]:
    return {pfx + "weight": 1.0 + 0.1 * torch.randn(n, generator=g), pfx + "bias": 0.1 * torch.randn(n, generator=g)}


def _resnet_hf_name(k: str) -> str:
    """torchvision resnet50 name -> transformers ResNetModel name."""
    if k.startswith("conv1."):
        return "embedder.embedder.convolution." + k[6:]
    if k.startswith("bn1."):
        return "embedder.embedder.normalization." + k[4:]
    m_ = re.match(r"layer(\d)\.(\d+)\.(conv|bn)(\d)\.(.*)", k)
    if m_:
        li, bi, kind, idx, rest = m_.groups()
        return f"encoder.stages.{int(li) - 1}.layers.{bi}.layer.{int(idx) - 1}.{'convolution' if kind == 'conv' else 'normalization'}.{rest}"
    m_ = re.match(r"layer(\d)\.(\d+)\.downsample\.(\d)\.(.*)", k)
    li, bi, idx, rest = m_.groups()
    return f"encoder.stages.{int(li) - 1}.layers.{bi}.shortcut.{'convolution' if idx == '0' else 'normalization'}.{rest}"


class TorchvisionResNet50(nn.Module):
    """What ``torchvision.models.resnet50`` returns, from transformers' ResNetModel: image [B, 3, H, W] -> 1000 logits.
    ``tv_state_dict()`` / ``load_tv_state()`` speak torchvision's key names."""

    def __init__(self, layers=(3, 4, 6, 3), width: int = 64, num_classes: int = 1000):
        super().__init__()
        from transformers import ResNetConfig, ResNetModel
        cfg = ResNetConfig(num_channels=3, embedding_size=width, hidden_sizes=[width * 4, width * 8, width * 16, width * 32],
                           depths=list(layers), layer_type="bottleneck", hidden_act="relu", downsample_in_bottleneck=False)
        self.body = ResNetModel(cfg)
        self.fc = nn.Linear(width * 32, num_classes)

    def forward(self, x):
        return self.fc(self.body(pixel_values=x).pooler_output.flatten(1))

    def _map(self, k):
        return k if k.startswith("fc.") else "body." + _resnet_hf_name(k)

    def load_tv_state(self, p: Dict[str, torch.Tensor]):
        sd = self.state_dict()
        mapped = {self._map(k): v.clone() for k, v in p.items()}
        missing = [k for k in mapped if k not in sd]
        assert not missing, missing[:4]
        extra = [k for k in sd if k not in mapped and "running" not in k and "num_batches" not in k]
        assert not extra, extra[:4]
        self.load_state_dict(mapped, strict=False)

    def tv_state_dict(self, names: Iterable[str]) -> Dict[str, torch.Tensor]:
        sd = self.state_dict()
        return {k: sd[self._map(k)] for k in names}


_VIT_RENAMES = (  # transformers 4.39.2 ViTModel names (oracle) -> names in the installed 5.x
    (r"encoder\.layer\.(\d+)\.attention\.attention\.query", r"layers.\1.attention.q_proj"),
    (r"encoder\.layer\.(\d+)\.attention\.attention\.key", r"layers.\1.attention.k_proj"),
    (r"encoder\.layer\.(\d+)\.attention\.attention\.value", r"layers.\1.attention.v_proj"),
    (r"encoder\.layer\.(\d+)\.attention\.output\.dense", r"layers.\1.attention.o_proj"),
    (r"encoder\.layer\.(\d+)\.intermediate\.dense", r"layers.\1.mlp.fc1"),
    (r"encoder\.layer\.(\d+)\.output\.dense", r"layers.\1.mlp.fc2"),
    (r"encoder\.layer\.(\d+)\.layernorm_", r"layers.\1.layernorm_"),
)


class TimmViT(nn.Module):
    """What ``timm.create_model("vit_*_patch16_224")`` + ``reset_classifier(0)`` computes: the final-LayerNorm'd cls token
    (token pooling, eps 1e-6), from transformers' ViTModel."""

    def __init__(self, image_size=224, patch=16, hidden=768, layers=12, heads=12, intermediate=3072, num_classes=1000):
        super().__init__()
        from transformers import ViTConfig, ViTModel
        vc = ViTConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=intermediate,
                       image_size=image_size, patch_size=patch, num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu",
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True)
        vc._attn_implementation = "eager"
        self.vit = ViTModel(vc, add_pooling_layer=False)
        self.head = nn.Linear(hidden, num_classes)
        self.num_features = hidden

    def reset_classifier(self, num_classes: int, global_pool=None):
        self.head = nn.Identity() if num_classes == 0 else nn.Linear(self.num_features, num_classes)

    def forward(self, x):
        return self.head(self.vit(pixel_values=x, return_dict=False)[0][:, 0])

    def _map(self, k: str) -> str:
        sd = self.vit.state_dict()
        if k in sd:
            return k
        for pat, rep in _VIT_RENAMES:
            k2 = re.sub(pat, rep, k)
            if k2 != k:
                return k2
        return k

    def load_oracle_state(self, p: Dict[str, torch.Tensor]):
        sd = self.vit.state_dict()
        mapped = {self._map(k): v.clone() for k, v in p.items()}
        missing = [k for k in mapped if k not in sd]
        assert not missing, missing[:4]
        extra = [k for k in sd if k not in mapped]
        assert not extra, extra[:4]
        self.vit.load_state_dict(mapped)

    def oracle_state_dict(self, names: Iterable[str]) -> Dict[str, torch.Tensor]:
        sd = self.vit.state_dict()
        return {k: sd[self._map(k)] for k in names}


def local_distilbert(vocab_size: int, layers: int, dropout: float, max_position: int = 512):
    from transformers import DistilBertConfig, DistilBertModel
    cfg = DistilBertConfig(vocab_size=vocab_size, dim=768, n_layers=layers, n_heads=12, hidden_dim=3072, max_position_embeddings=max_position,
                           dropout=dropout, attention_dropout=dropout, sinusoidal_pos_embds=False, pad_token_id=0)
    cfg._attn_implementation = "eager"
    return DistilBertModel(cfg)


def local_bert(vocab_size: int, layers: int, dropout: float, max_position: int = 512, hidden: int = 768, heads: int = 12,
               intermediate: int = 3072):
    from transformers import BertConfig, BertModel
    bc = BertConfig(vocab_size=vocab_size, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                    intermediate_size=intermediate, max_position_embeddings=max_position, type_vocab_size=2, layer_norm_eps=1e-12,
                    hidden_act="gelu", hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, pad_token_id=0)
    bc._attn_implementation = "eager"
    return BertModel(bc, add_pooling_layer=False)


def load_text_state(module: nn.Module, state: Dict[str, torch.Tensor]):
    sd = module.state_dict()
    missing = [k for k in state if k not in sd]
    assert not missing, missing[:4]
    extra = [k for k in sd if k not in state and not k.endswith("position_ids") and "token_type_ids" not in k]
    assert not extra, extra[:4]
    module.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)


def focal_standin(inputs, targets, alpha: float = 0.25, gamma: float = 2, reduction: str = "none"):
    """``torchvision.ops.sigmoid_focal_loss(inputs, targets, alpha, gamma, reduction)`` through transformers' DETR helper
    (``loss.mean(1).sum() / num_boxes``): with inputs [B] -> [B, 1] and num_boxes = B that is reduction='mean'."""
    from transformers.loss.loss_for_object_detection import sigmoid_focal_loss as hf_focal
    assert reduction == "mean", "only the reduction the reference passes is restated"
    return hf_focal(inputs[:, None], targets[:, None], num_boxes=inputs.numel(), alpha=alpha, gamma=gamma)


SAMPLE = 6


def sample_index(n: int) -> torch.Tensor:
    return (torch.arange(SAMPLE, dtype=torch.int64) * max(n - 1, 0)) // (SAMPLE - 1)


def sample_of(t: torch.Tensor) -> np.ndarray:
    f = t.detach().reshape(-1)
    return f[sample_index(f.numel())].float().cpu().numpy().astype(np.float32)


class ListLoader:
    """A DataLoader stand-in over ready batches (dicts): what the reference's loops need is iteration, ``len()`` and
    ``.dataset`` with a length."""

    def __init__(self, batches: List[dict]):
        self.batches = batches
        self.dataset = range(sum(len(b["id"]) for b in batches))

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


# ----------------------------------------------------------------------------------------------------------------------
# whole-model states under the REFERENCE's state_dict key names
# ----------------------------------------------------------------------------------------------------------------------
def organizers_state(vocab_size: int, text_layers: int, resnet_layers=(3, 4, 6, 3), seed: int = 31, num_classes: int = 2) -> Dict[str, torch.Tensor]:
    """``MultimodalClassifier(num_classes).state_dict()`` of Multimodal_example_task2C.txt:152-170 (parameters only): ``bert.*``
    (DistilBertModel names), ``resnet.*`` (torchvision resnet50 names), ``bert_fc / resnet_fc / fusion_fc / output_fc``."""
    from . import resnet_oracle as R
    out = {"bert." + k: v for k, v in seeded_text_state(vocab_size, text_layers, seed, type_vocab=0, naming="distilbert").items()}
    out.update({"resnet." + k: v for k, v in R.resnet_init(resnet_layers, 64, 1000, seed + 1).items()})
    g = torch.Generator().manual_seed(seed + 2)
    for name, (o, i) in (("bert_fc", (512, 768)), ("resnet_fc", (512, 1000)), ("fusion_fc", (512, 1024)), ("output_fc", (num_classes, 512))):
        out[name + ".weight"], out[name + ".bias"] = seeded_linear(o, i, g)
    return out


KEVIN = dict(seq_len=128, batch=6, text_layers=4, caption_layers=2, vit=dict(image_size=224, patch=16, hidden=512, layers=4, heads=8, intermediate=2048),
             proj=512, seed=41, lr=1e-5, aug_seed=77)


def kevin_state(text_vocab: int, caption_vocab: int, cfg=KEVIN) -> Dict[str, torch.Tensor]:
    """Parameters of ``MultimodalClassifier(fusion_method="concatenation")`` (Multimodal_example_task2C.py:587-643) under the
    reference's module paths: ``text_model.model.*`` / ``caption_text_model.model.*`` (BertModel names), ``image_model.image_model.*``
    (the ViT in the oracle's transformers-4.39.2 naming -- ``to_timm_names`` gives timm's own), ``image_model.fine_tune.{0,3}``,
    ``text_fc / caption_text_fc.{0,1}``, ``fusion_layer.attention_layer.{0,1}``, ``fusion_layer.reduce.{0,1}``, ``output_fc.{0,1}``."""
    s, P, v = cfg["seed"], cfg["proj"], cfg["vit"]
    out = {"text_model.model." + k: t for k, t in seeded_text_state(text_vocab, cfg["text_layers"], s).items()}
    out.update({"caption_text_model.model." + k: t for k, t in seeded_text_state(caption_vocab, cfg["caption_layers"], s + 1).items()})
    out.update({"image_model.image_model." + k: t for k, t in seeded_vit_state(s + 2, **v).items()})
    g = torch.Generator().manual_seed(s + 3)

    def lin(name, o, i):
        out[name + ".weight"], out[name + ".bias"] = seeded_linear(o, i, g)

    lin("image_model.fine_tune.0", 512, v["hidden"]); lin("image_model.fine_tune.3", 512, 512)
    lin("text_fc.0", P, 768); out.update(seeded_bn(P, g, "text_fc.1."))
    lin("caption_text_fc.0", P, 768); out.update(seeded_bn(P, g, "caption_text_fc.1."))
    lin("fusion_layer.attention_layer.0", 3 * P, 3 * P); out.update(seeded_bn(3 * P, g, "fusion_layer.attention_layer.1."))
    lin("fusion_layer.reduce.0", P, 3 * P); out.update(seeded_bn(P, g, "fusion_layer.reduce.1."))
    lin("output_fc.0", 1, P); out.update(seeded_bn(1, g, "output_fc.1."))
    return out


def to_timm_names(vit_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Oracle / transformers-4.39.2 ViTModel names -> timm 0.9.16 VisionTransformer names (q, k, v fused into ``attn.qkv``)."""
    out = {"cls_token": vit_state["embeddings.cls_token"], "pos_embed": vit_state["embeddings.position_embeddings"],
           "patch_embed.proj.weight": vit_state["embeddings.patch_embeddings.projection.weight"],
           "patch_embed.proj.bias": vit_state["embeddings.patch_embeddings.projection.bias"],
           "norm.weight": vit_state["layernorm.weight"], "norm.bias": vit_state["layernorm.bias"]}
    n = 1 + max(int(k.split(".")[2]) for k in vit_state if k.startswith("encoder.layer."))
    for i in range(n):
        L, B = f"encoder.layer.{i}.", f"blocks.{i}."
        for wb in ("weight", "bias"):
            out[B + "norm1." + wb] = vit_state[L + "layernorm_before." + wb]
            out[B + "norm2." + wb] = vit_state[L + "layernorm_after." + wb]
            out[B + "attn.qkv." + wb] = torch.cat([vit_state[L + f"attention.attention.{q}." + wb] for q in ("query", "key", "value")], 0)
            out[B + "attn.proj." + wb] = vit_state[L + "attention.output.dense." + wb]
            out[B + "mlp.fc1." + wb] = vit_state[L + "intermediate.dense." + wb]
            out[B + "mlp.fc2." + wb] = vit_state[L + "output.dense." + wb]
    return out


def kevin_transform():
    """Multimodal_example_task2C.py:222-235."""
    T = transforms
    return T.Compose([T.Resize((224, 224)), T.RandomHorizontalFlip(), T.ColorJitter(brightness=0.1, contrast=0.1, saturation=0.1, hue=0.1),
                      T.RandomRotation(degrees=15), T.ToTensor(), T.Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))])


def kevin_images(root: str, aug_seed: int) -> torch.Tensor:
    """The 24 image tensors Kevin's Dataset yields when its items are read once, in index order, after ``AUG_GEN.manual_seed``."""
    from PIL import Image
    tf = kevin_transform()
    AUG_GEN.manual_seed(aug_seed)
    return torch.stack([tf(Image.open(os.path.join(root, r["img_path"])).convert("RGB")) for r in records24()])


def bn_buffers_from_fixture(z, prefix: str = "") -> Dict[str, torch.Tensor]:
    """``init_bn_names / init_bn_running_mean / init_bn_running_var`` of a reference-run fixture -> state_dict entries
    ``<prefix><name>.running_mean`` / ``.running_var`` (+ ``num_batches_tracked`` = 0)."""
    out, pos = {}, 0
    mean, var = z["init_bn_running_mean"], z["init_bn_running_var"]
    sizes = z["init_bn_sizes"]
    for name, n in zip(z["init_bn_names"], sizes):
        n = int(n)
        out[f"{prefix}{name}.running_mean"] = torch.from_numpy(mean[pos:pos + n].copy())
        out[f"{prefix}{name}.running_var"] = torch.from_numpy(var[pos:pos + n].copy())
        out[f"{prefix}{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        pos += n
    return out


def kevin_head_cpu(state: Dict[str, torch.Tensor], t: torch.Tensor, vi: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """Kevin's head (Multimodal_example_task2C.py:599-643,666-685) in fp32 torch on the CPU, train-mode BatchNorm (batch statistics),
    dropout off: pooled text / image / caption features -> the [B] outputs.  Restated for the tests (checked against the reference's
    own forward by tests/test_reference_run.py)."""
    import torch.nn.functional as F

    def lin(name, x):
        return F.linear(x, state[name + ".weight"], state[name + ".bias"])

    def bn(name, x):
        return F.batch_norm(x, None, None, state[name + ".weight"], state[name + ".bias"], True, 0.1, 1e-5)

    to = F.relu(bn("text_fc.1", lin("text_fc.0", t)))
    co = F.relu(bn("caption_text_fc.1", lin("caption_text_fc.0", c)))
    io = lin("image_model.fine_tune.3", F.relu(lin("image_model.fine_tune.0", vi)))
    cat = torch.cat((to, io, co), dim=1)
    att = torch.softmax(F.relu(bn("fusion_layer.attention_layer.1", lin("fusion_layer.attention_layer.0", cat))), dim=1)
    red = F.relu(bn("fusion_layer.reduce.1", lin("fusion_layer.reduce.0", att * cat)))
    return bn("output_fc.1", lin("output_fc.0", red)).squeeze(1)
