"""Small helpers of the training harness (reference trainers/train_helpers.py: `cycle`; the wandb / torchvision image
logging of the reference is replaced by a JSON-lines logger with the same call shape -- wandb is not installable
offline and logging is outside the accelerated path)."""
import json
import os
import time


def cycle(dl):
    """train_helpers.py: endless iterator over a DataLoader."""
    while True:
        for data in dl:
            yield data


def delete_if_exists(path):
    if path and os.path.exists(path):
        os.remove(path)


class RunLogger:
    """wandb-shaped shim: init / log / save / finish, writing one JSON object per log call."""

    def __init__(self, directory, project, config, run_id=None, enabled=True):
        self.id = run_id or time.strftime("%Y%m%d%H%M%S") + f"{os.getpid() % 10000:04d}"
        self.enabled = enabled
        self.path = None
        if enabled:
            os.makedirs(directory, exist_ok=True)
            self.path = os.path.join(directory, f"{project}_{self.id}.jsonl")
            with open(self.path, "a") as f:
                f.write(json.dumps({"event": "init", "config": {k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v))
                                                                for k, v in config.items()}}) + "\n")
        self._pending = {}

    def log(self, values, commit=True):
        self._pending.update({k: float(v) for k, v in values.items()})
        if commit and self.enabled:
            with open(self.path, "a") as f:
                f.write(json.dumps(self._pending) + "\n")
            self._pending = {}

    def save(self, path, policy=None):
        pass

    def finish(self):
        if self._pending:
            self.log({}, commit=True)
