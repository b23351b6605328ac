from .layers import Linear, LayerNorm, MultiheadAttention, Dropout, MLPHead  # noqa: F401
from .resnet import ResNet, BasicBlock, Bottleneck, Stem, resnet18, resnet34, resnet50  # noqa: F401
from .bert import BertConfig, BertModel  # noqa: F401
