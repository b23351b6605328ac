"""Fusion operators, heads, gate and tabular encoder -- same public names as the reference's
`modules` package (modules/__init__.py:1-29), implemented on the hamspine HIP kernels."""
from .fusion_blocks import (  # noqa: F401
    BasicTransformerBlock,
    BilinearFusionModule,
    ConcatFusionModule,
    CrossAttentionBlock,
    FusionModule,
    HadamardFusionModule,
    MultiScaleFusionModule,
    SSMFusionModule,
    VMambaFusionModule,
    WeightedConcatFusionModule,
)
from .gating import DualExpertGate  # noqa: F401
from .heads import AttentionPoolingClassifier, ResidualClassifier, build_kan_head  # noqa: F401
from .tabular import TabularEncoder  # noqa: F401

__all__ = [
    "FusionModule", "ConcatFusionModule", "MultiScaleFusionModule", "WeightedConcatFusionModule",
    "HadamardFusionModule", "BilinearFusionModule", "SSMFusionModule", "VMambaFusionModule",
    "ResidualClassifier", "AttentionPoolingClassifier", "build_kan_head", "DualExpertGate", "TabularEncoder",
]
