"""evaluations/ of the reference."""
from .evaluation import evaluate, eval_step, prepare_dataset  # noqa: F401
