"""Host-side pieces of the replay path that need no GPU."""
import random

import numpy as np


def test_vectorised_uniform_draws_are_the_python_stream():
    """PrioritizedRB.draw_uniforms lends CPython's MT19937 state to numpy for batches >= 1024: the
    draws and the generator state afterwards must equal B calls of random.random() (memory.py:278)."""
    from sam_tf_amd.memory import PrioritizedRB
    draw = PrioritizedRB.draw_uniforms
    for seed, B in ((0, 1024), (7, 8192), (123, 5000)):
        random.seed(seed)
        random.random()                      # odd position inside the 624-word block
        want = np.array([random.random() for _ in range(B)])
        after = [random.random() for _ in range(3)]
        random.seed(seed)
        random.random()
        got = draw(None, B)
        assert got.dtype == np.float64 and np.array_equal(got, want)
        assert [random.random() for _ in range(3)] == after
    random.seed(1)
    small = draw(None, 16)                   # the plain loop
    random.seed(1)
    assert np.array_equal(small, [random.random() for _ in range(16)])
