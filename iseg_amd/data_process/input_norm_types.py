"""data_process/input_norm_types.py of the reference."""
from enum import Enum


class InputNormTypes(Enum):
    NONE = 0
    ZERO_MEAN = 1
    KERAS = 2
    KERAS_SCALE = 3
