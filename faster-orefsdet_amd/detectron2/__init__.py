"""Thin `detectron2`-named namespace (NEW code, not the reference's vendored tree): just the names
`fsod_train_net.py` and `fewx` import, backed by the MI355X HIP path (orehip / libore_hip.so).
See SURVEY.md section 8b for the list of names this surface must expose."""
__version__ = "0.5+orehip"
