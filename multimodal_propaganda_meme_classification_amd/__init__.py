"""MI355X-native dual-encoder fine-tuning path for Subtask-2C meme classification.

Host side is Python (the reference is Python); all arithmetic runs in hand-written HIP kernels
for gfx950 behind the C ABI in include/memehip.h (libmemehip.so).  No CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import MemehipError  # noqa: F401
from .config import ImageConfig, Layout, ModelConfig, TextConfig  # noqa: F401
from .model import (Adam, BatchNorm1d, CrossEntropyLoss, GradScaler, GraphedStep, MultimodalClassifier, SigmoidFocalLoss,  # noqa: F401
                    TextEncoder, flatten_parameters, get_linear_schedule_with_warmup)
from . import fused  # noqa: F401
from .heads import (MCA3, ConcatAttention3, FineTuneMLP, KevinMultimodalClassifier, LinearBNReLU,  # noqa: F401
                    OrganizersMultimodalClassifier, SequencePooling, TextClassifier, TrainerModel)
from .data import HashTokenizer, MultimodalDataset, id2l, l2id, normalize_images, read_data  # noqa: F401
from .train import evaluate, test, train  # noqa: F401
from . import kevin  # noqa: F401
from .kevin import KevinMultimodalDataset, kevin_collate  # noqa: F401
from .resnet import Bottleneck, ResNet50, ResNetClassifier  # noqa: F401
from .convnext import CNBlock, ConvNeXtTiny  # noqa: F401
from .features import dump_features, get_features  # noqa: F401
