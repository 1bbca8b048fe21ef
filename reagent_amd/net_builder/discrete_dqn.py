"""reagent/net_builder/discrete_dqn/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["discrete_dqn"])
__all__ = sorted(_B["discrete_dqn"])
