"""Preprocessor with the surface of reagent/preprocessing/preprocessor.py:23-599, executed by one
rg_normalize_dense launch (per-column op-code table instead of split/per-type tensor ops/cat).

Input columns are in ``sorted_features`` order (by feature type in FEATURE_TYPES order, then by
feature id — preprocessor.py:527-545), the output has one column per feature except ENUM features,
which expand to ``len(possible_values)`` columns.  The reference's training-mode range check
(`_check_preprocessing_output`, :576-599: batch.min()/.max() + .item() host syncs per type group)
is not reproduced: it only raises on out-of-range outputs of already-clamped types.
"""
import ctypes
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.nn import Module

from .. import _lib as L
from .. import ops

FEATURE_TYPES = ("BINARY", "PROBABILITY", "CONTINUOUS", "BOXCOX", "ENUM", "QUANTILE", "CONTINUOUS_ACTION",
                 "DISCRETE_ACTION", "DO_NOT_PREPROCESS", "CLIP_LOG")  # identify_types.py:19-30
_OP = {t: i for i, t in enumerate(FEATURE_TYPES)}
EPS = 1e-6


class Preprocessor(Module):
    def __init__(self, normalization_parameters: Dict[int, object], use_gpu: Optional[bool] = None,
                 device: Optional[torch.device] = None) -> None:
        super().__init__()
        self.normalization_parameters = normalization_parameters
        assert isinstance(list(normalization_parameters.keys())[0], int), "Normalization Parameters need to be int"
        self.feature_id_to_index, self.sorted_features, self.sorted_feature_boundaries = (
            self._sort_features_by_normalization()
        )
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        cols: List[Tuple[int, int, float, float, float, float]] = []
        quantiles: List[float] = []
        for j, f in enumerate(self.sorted_features):
            p = normalization_parameters[f]
            t = p.feature_type
            if t == "CONTINUOUS":
                cols.append((_OP[t], j, p.mean, p.stddev, 0.0, 0.0))
            elif t == "BOXCOX":
                assert abs(p.boxcox_lambda) > 1e-6, "Invalid value for boxcox lambda: " + str(p.boxcox_lambda)
                cols.append((_OP[t], j, p.boxcox_shift, p.boxcox_lambda, p.mean, p.stddev))
            elif t == "ENUM":
                for v in p.possible_values:
                    cols.append((_OP[t], j, float(v), 0.0, 0.0, 0.0))
            elif t == "QUANTILE":
                cols.append((_OP[t], j, float(len(quantiles)), float(len(p.quantiles)), 0.0, 0.0))
                quantiles += [float(q) for q in p.quantiles]
            elif t == "CONTINUOUS_ACTION":
                # same fp32 torch arithmetic as _create_parameters_CONTINUOUS_ACTION (:246-268)
                scaling = ((torch.ones(1) - EPS) * 2 / torch.tensor([p.max_value - p.min_value])).item()
                min_training = (torch.ones(1) * -1 + EPS).item()
                cols.append((_OP[t], j, p.min_value, scaling, min_training, 0.0))
            elif t in _OP:
                cols.append((_OP[t], j, 0.0, 0.0, 0.0, 0.0))
            else:
                raise ValueError(f"unknown feature type {t}")
        self.num_output_features = len(cols)
        # 1:1 table (no ENUM expansion): usable as the normalize-on-gather epilogue of the replay buffer
        self.elementwise = len(cols) == len(self.sorted_features) and all(c[1] == i for i, c in enumerate(cols))
        table = (L.NormCol * len(cols))()
        for i, c in enumerate(cols):
            table[i].op, table[i].in_col = c[0], c[1]
            table[i].p0, table[i].p1, table[i].p2, table[i].p3 = (np.float32(c[2]), np.float32(c[3]),
                                                                np.float32(c[4]), np.float32(c[5]))
        raw = np.frombuffer(bytes(table), dtype=np.uint8).copy()
        self.register_buffer("_col_table", torch.from_numpy(raw).to(self.device), persistent=False)
        q = torch.tensor(quantiles if quantiles else [0.0], dtype=torch.float32)
        self.register_buffer("_quantiles", q.to(self.device), persistent=False)

    def input_prototype(self) -> Tuple[torch.Tensor, torch.Tensor]:
        n = len(self.normalization_parameters)
        return (torch.randn(1, n, device=self.device),
                torch.ones(1, n, dtype=torch.uint8, device=self.device))

    @torch.no_grad()
    def forward(self, input: torch.Tensor, input_presence_byte: torch.Tensor) -> torch.Tensor:
        assert input.shape == input_presence_byte.shape, f"{input.shape} != {input_presence_byte.shape}"
        x = input if input.dtype == torch.float32 else input.float()
        x = x if x.stride(-1) == 1 else x.contiguous()
        pres = input_presence_byte
        if pres.dtype == torch.bool:
            pres = pres.view(torch.uint8)
        elif pres.dtype != torch.uint8:
            pres = pres.to(torch.uint8)
        pres = pres if pres.stride(-1) == 1 else pres.contiguous()
        out = torch.empty(x.shape[0], self.num_output_features, dtype=torch.float32, device=x.device)
        ops.normalize_dense(x, pres, self._col_table, self.num_output_features, self._quantiles, out)
        return out

    def _sort_features_by_normalization(self):
        feature_id_to_index, sorted_features, feature_starts = {}, [], []
        for feature_type in FEATURE_TYPES:
            feature_starts.append(len(sorted_features))
            for feature in sorted(self.normalization_parameters.keys()):
                if self.normalization_parameters[feature].feature_type == feature_type:
                    feature_id_to_index[feature] = len(sorted_features)
                    sorted_features.append(feature)
        return feature_id_to_index, sorted_features, feature_starts
