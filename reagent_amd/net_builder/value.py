"""reagent/net_builder/value/*: the builders of this family under the reference's class names"""
from . import BUILDERS as _B

globals().update(_B["value"])
__all__ = sorted(_B["value"])
