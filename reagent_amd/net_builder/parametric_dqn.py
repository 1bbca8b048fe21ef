"""reagent/net_builder/parametric_dqn/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["parametric_dqn"])
__all__ = sorted(_B["parametric_dqn"])
