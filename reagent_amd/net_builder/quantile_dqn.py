"""reagent/net_builder/quantile_dqn/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["quantile_dqn"])
__all__ = sorted(_B["quantile_dqn"])
