"""Utilities for LaTok tokenization -- mirror of reference latok/core/latok_utils.py (same public names)."""
from dataclasses import dataclass

import numpy as np

from ..latok import _gen_block_mask, _gen_parse_matrix


def gen_parse_matrix(text: str) -> np.ndarray:
    """Feature matrix of a string: one row per character, one column per feature of offsets.py
    (reference latok_utils.py:10-15)."""
    return _gen_parse_matrix(text)


def gen_block_mask(a1: np.ndarray, a2: np.ndarray) -> np.ndarray:
    """Mask of ones with zeros between a2's ones wherever a1 has a one, a2's end points counting as ones
    (reference latok_utils.py:18-24)."""
    return _gen_block_mask(a1, a2)


def build_combo_matrix(idx_lists):
    """Rule table from a list of index lists: every inner list is a conjunction of feature columns ("and" =
    multiply), the outer list is their disjunction ("or" = add); ragged rows are padded with -1
    (reference latok_utils.py:27-56)."""
    width = max(len(row) for row in idx_lists)
    combo = np.full((len(idx_lists), width), -1, dtype=np.int8)
    for r, row in enumerate(idx_lists):
        combo[r, :len(row)] = row
    return combo


# Names of the features of offsets.py, in column order (reference latok_utils.py:60-86)
FEATURE_NAMES = [
    'Alpha', 'AlphaNum', 'Num', 'Lower', 'Upper', 'Space', 'Symbol', 'Twitter', '@', ':', '/', '.',
    'Prev_Alpha', 'Next_Alpha', 'Prev_AlphaNum', 'Next_AlphaNum', 'Prev_Lower', 'Next_Lower', 'Prev_Space',
    'Next_Space', 'Prev_Symbol', 'Next_@', 'Next_/', 'After_Next_Alpha', 'After_Next_/',
]

NUM_FEATURES = len(FEATURE_NAMES)


@dataclass
class LaToken:
    """A token with its span and feature vector (reference latok_utils.py:92-116)."""
    text: str
    start_idx: int
    end_idx: int
    features: np.ndarray

    def weight(self, weighting=None):
        """Sum of the (optionally weighted) features."""
        return np.sum((self.features * weighting) if weighting else self.features)

    def feature_weights(self):
        """Non-zero feature names mapped to their weights."""
        return {FEATURE_NAMES[i]: self.features[i] for i in np.nonzero(self.features)[0]}
