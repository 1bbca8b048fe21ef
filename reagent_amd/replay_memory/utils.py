"""replay_buffer_to_pre_timeline_df of reagent/replay_memory/utils.py:27-82: the whole buffer as the
pre-timeline table (one row per transition, sparse feature maps) that the reference uploads for its
timeline operator.  The rows are fetched by one HIP gather of `size` sampled transitions; the table
itself is host-side pandas, exactly the reference's columns and Python types.
"""
from typing import Dict, List

import numpy as np
import pandas as pd

DEFAULT_DS = "2019-01-01"


def _dense_to_sparse(dense: np.ndarray) -> List[Dict[int, float]]:
    assert len(dense.shape) == 2, f"dense shape is {dense.shape}"
    return [dict(enumerate(row)) for row in dense.tolist()]


def replay_buffer_to_pre_timeline_df(is_discrete_action: bool, replay_buffer) -> pd.DataFrame:
    n = replay_buffer.size
    batch = replay_buffer.sample_transition_batch(batch_size=n)

    def host(name):
        return getattr(batch, name).cpu().numpy()

    terminal = host("terminal").squeeze(1).tolist()
    action_arr = host("action")
    assert len(action_arr.shape) == 2
    possible_actions_mask = getattr(batch, "possible_actions_mask", None)
    possible_actions = getattr(batch, "possible_actions", None)
    if is_discrete_action:
        assert action_arr.shape[1] == 1, f"discrete action batch with shape {action_arr.shape}"
        action = [str(a) for a in action_arr[:, 0].tolist()]  # action names are strings
        # the action space is taken to be what the buffer has seen (utils.py:47-56)
        unique_actions = np.unique(action_arr)
        names = [str(a) for a in unique_actions]
        possible_actions_mask = [[] if t else [1] * len(names) for t in terminal]
        possible_actions = [[] if t else list(names) for t in terminal]
    else:
        action = _dense_to_sparse(action_arr)  # map<str, double> of a Box action
    reward = host("reward").squeeze(1).tolist()
    rows = {
        "ds": [DEFAULT_DS] * n,
        "state_features": _dense_to_sparse(host("state")),
        "action": action,
        "mdp_id": [str(m) for m in host("mdp_id").flatten().tolist()],
        "sequence_number": host("sequence_number").squeeze(1).tolist(),
        "action_probability": np.exp(host("log_prob").squeeze(1)).tolist(),
        "reward": reward,
        "metrics": [{"reward": r} for r in reward],
    }
    if possible_actions_mask is not None:
        rows["possible_actions_mask"] = possible_actions_mask
    if possible_actions is not None:
        rows["possible_actions"] = possible_actions
    return pd.DataFrame.from_dict(rows)
