"""reagent/net_builder/categorical_dqn/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["categorical_dqn"])
__all__ = sorted(_B["categorical_dqn"])
