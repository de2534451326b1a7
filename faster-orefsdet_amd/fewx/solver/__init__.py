from .build import (FlatBucket, FlatDataParallel, FlatSGD, WarmupMultiStepLR, build_lr_scheduler, build_optimizer, get_bucket,
                    param_groups_like_reference, warmup_factor_at_iter)
from .graphed_step import GraphedTrainStep  # noqa: E402,F401
