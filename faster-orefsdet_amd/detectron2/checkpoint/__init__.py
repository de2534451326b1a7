"""DetectionCheckpointer (d2z:checkpoint/detection_checkpoint.py, fvcore Checkpointer): `.pth` files hold {"model": state_dict,
"optimizer": ..., "scheduler": ..., "iteration": n}; reference checkpoints load by name."""
import os

import torch


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=True, **checkpointables):
        self.model, self.save_dir, self.save_to_disk = model, save_dir, save_to_disk
        self.checkpointables = dict(checkpointables)

    def _last_file(self):
        return os.path.join(self.save_dir, "last_checkpoint")

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(self._last_file())

    def get_checkpoint_file(self):
        with open(self._last_file()) as f:
            return os.path.join(self.save_dir, f.read().strip())

    def save(self, name, **extra):
        if not self.save_dir or not self.save_to_disk:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        data = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(extra)
        fn = name + ".pth"
        torch.save(data, os.path.join(self.save_dir, fn))
        with open(self._last_file(), "w") as f:
            f.write(fn)

    def load(self, path, checkpointables=None):
        if not path:
            return {}
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
        with torch.no_grad():                                   # copy_ keeps parameters that live in the flat bucket in place
            own = self.model.state_dict()
            missing = [k for k in own if k not in sd]
            unexpected = [k for k in sd if k not in own]
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(torch.as_tensor(v).to(own[k].dtype))
        for k in (self.checkpointables if checkpointables is None else checkpointables):
            if k in ck and k in self.checkpointables:
                self.checkpointables[k].load_state_dict(ck[k])
        try:
            from orehip import autograd as A
            A.weights_changed()
        except Exception:
            pass
        ck["__missing__"], ck["__unexpected__"] = missing, unexpected
        return ck

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])
