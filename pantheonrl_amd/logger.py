"""Key/value logger with the small surface PantheonRL uses from `stable_baselines3.common.logger`:
`configure_logger(verbose, tensorboard_log, tb_log_name)`, `Logger.record(key, value, exclude=)`, `Logger.dump(step)`
(reference pantheonrl/common/agents.py:102-107,134-153).  Output formats: an stdout table when verbose, and an
append-only JSON-lines file under `<tensorboard_log>/<tb_log_name>_<n>/progress.jsonl` (tensorboard itself is not
installed here; the directory naming follows SB3's `<name>_<run id>` rule so existing tooling finds the runs).
"""
from __future__ import annotations

import json
import os
import sys
from typing import Any, Dict, Optional


def safe_mean(values) -> float:
    """mean that returns nan for an empty sequence (stable_baselines3.common.utils.safe_mean; agents.py:145-147)."""
    values = list(values)
    return float("nan") if len(values) == 0 else float(sum(values) / len(values))


class Logger:
    def __init__(self, folder: Optional[str] = None, stdout: bool = False):
        self.folder, self.stdout = folder, stdout
        self.name_to_value: Dict[str, Any] = {}
        self.name_to_excluded: Dict[str, Any] = {}
        self.history = []  # every dumped record, newest last (tests and callers can read it back)
        if folder is not None:
            os.makedirs(folder, exist_ok=True)

    def record(self, key: str, value: Any, exclude=None) -> None:
        self.name_to_value[key] = value
        self.name_to_excluded[key] = exclude

    def dump(self, step: int = 0) -> None:
        rec = dict(self.name_to_value)
        rec["_step"] = step
        self.history.append(rec)
        if self.stdout and rec:
            width = max(len(k) for k in rec)
            bar = "-" * (width + 18)
            lines = [bar] + [f"| {k:<{width}} | {_fmt(v):>11} |" for k, v in rec.items() if k != "_step"] + [bar]
            sys.stdout.write("\n".join(lines) + "\n")
            sys.stdout.flush()
        if self.folder is not None:
            keep = {k: _json(v) for k, v in rec.items()
                    if not _excluded(self.name_to_excluded.get(k), "tensorboard")}
            with open(os.path.join(self.folder, "progress.jsonl"), "a") as fh:
                fh.write(json.dumps(keep) + "\n")
        self.name_to_value.clear()
        self.name_to_excluded.clear()


def _excluded(spec, fmt: str) -> bool:
    if spec is None:
        return False
    return fmt == spec if isinstance(spec, str) else fmt in spec


def _fmt(v) -> str:
    if isinstance(v, float):
        return f"{v:.4g}"
    return str(v)[:11]


def _json(v):
    try:
        json.dumps(v)
        return v
    except TypeError:
        return float(v) if hasattr(v, "__float__") else str(v)


def configure_logger(verbose: int = 0, tensorboard_log: Optional[str] = None, tb_log_name: str = "") -> Logger:
    folder = None
    if tensorboard_log is not None:
        run = 1
        while os.path.exists(os.path.join(tensorboard_log, f"{tb_log_name}_{run}")):
            run += 1
        folder = os.path.join(tensorboard_log, f"{tb_log_name}_{run}")
    return Logger(folder=folder, stdout=verbose >= 1)
