import random

import numpy as np
import torch

TORCH_VERSION = tuple(int(x) for x in torch.__version__.split(".")[:2])


def seed_all_rng(seed=None):
    if seed is None or seed < 0:
        import os
        import time
        seed = (os.getpid() + int(time.time() * 1e6)) % (2 ** 31)
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
    return seed
