import logging
import sys
from collections import Counter

_LOG_COUNTER = Counter()


def setup_logger(output=None, distributed_rank=0, *, color=True, name="detectron2", abbrev_name=None):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    if distributed_rank == 0 and not logger.handlers:
        ch = logging.StreamHandler(stream=sys.stdout)
        ch.setLevel(logging.DEBUG)
        ch.setFormatter(logging.Formatter("[%(asctime)s %(name)s]: %(message)s", datefmt="%m/%d %H:%M:%S"))
        logger.addHandler(ch)
    if output is not None and distributed_rank == 0:
        import os
        fn = output if output.endswith((".txt", ".log")) else os.path.join(output, "log.txt")
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        fh = logging.FileHandler(fn)
        fh.setFormatter(logging.Formatter("[%(asctime)s] %(name)s %(levelname)s: %(message)s"))
        logger.addHandler(fh)
    return logger


def log_first_n(lvl, msg, n=1, *, name=None, key="caller"):
    _LOG_COUNTER[msg] += 1
    if _LOG_COUNTER[msg] <= n:
        logging.getLogger(name or "detectron2").log(lvl, msg)
