"""d2z:solver names the reference imports; the implementation is fewx.solver (flat-bucket SGD on the HIP kernel)."""
from fewx.solver.build import WarmupMultiStepLR, build_lr_scheduler, build_optimizer, warmup_factor_at_iter  # noqa: F401
