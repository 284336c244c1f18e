"""
Transition module factory.  Reference: models/Blocks/transition_models.py:12-39.
"""

import torch.nn as nn

from .attention import TransformerBlock

__all__ = ["get_transition_module"]


def get_transition_module(model_name, **kwargs):
    """ '' / None -> identity; 'TransformerBlock' -> POST-norm block (transition_models.py:24-29) """
    slot_dim = kwargs.pop("slot_dim")
    if model_name in [None, ""]:
        return nn.Identity()
    if model_name == "TransformerBlock":
        return TransformerBlock(embed_dim=slot_dim, pre_norm=False, **kwargs)
    raise ValueError(f"UPSI, {model_name = } was not a recognized transition module...")
