"""reagent/net_builder/continuous_actor/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["continuous_actor"])
__all__ = sorted(_B["continuous_actor"])
