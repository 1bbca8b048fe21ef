from .predictor_wrapper import (  # noqa: F401
    DiscreteDqnPredictorWrapper,
    DiscreteDqnWithPreprocessor,
    ServingFeatureData,
)
