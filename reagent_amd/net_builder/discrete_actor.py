"""reagent/net_builder/discrete_actor/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["discrete_actor"])
__all__ = sorted(_B["discrete_actor"])
