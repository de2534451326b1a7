"""Minimal EventStorage (d2z:utils/events.py): scalars per iteration + a context-managed current storage."""
from collections import defaultdict

_CURRENT = []


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._latest = {}
        self._history = defaultdict(list)

    def put_scalar(self, name, value, smoothing_hint=True, cur_iter=None):
        v, it = float(value), self.iter if cur_iter is None else cur_iter
        self._latest[name] = (v, it)
        self._history[name].append((v, it))

    def put_scalars(self, *, smoothing_hint=True, cur_iter=None, **kw):
        for k, v in kw.items():
            self.put_scalar(k, v, cur_iter=cur_iter)

    def put_image(self, name, img):
        pass

    def latest(self):
        return self._latest

    def history(self, name):
        return self._history[name]

    def step(self):
        self.iter += 1

    def __enter__(self):
        _CURRENT.append(self)
        return self

    def __exit__(self, *a):
        assert _CURRENT[-1] is self
        _CURRENT.pop()


def get_event_storage():
    assert _CURRENT, "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT[-1]


def has_event_storage():
    return bool(_CURRENT)
