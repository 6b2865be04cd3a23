"""oracle/ -- CPU restatement of the reference's TensorFlow/Keras semantics.  TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package, and only as the
checker.  PARITY UNPINNED (see oracle/tf_ops.py header): TensorFlow is unavailable here and the reference has no
golden vectors for this path, so the oracle is pinned by closed-form known-answer tests and a second,
independent numpy-loop implementation only.
"""
