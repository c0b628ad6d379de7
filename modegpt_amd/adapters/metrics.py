"""Run metrics store: one dict per run keyed by its start time, persisted to ./metrics/metrics.json plus one
./metrics/jsons/<run>--<note>.json per run (reference behaviour: src/adapters/model_adapter.py:85-94,137-182)."""
from __future__ import annotations

import json
import os
from datetime import datetime
from typing import Optional

ALL_RUNS: dict = {}


def load(path="./metrics/metrics.json") -> None:
    if not ALL_RUNS and os.path.exists(path):
        with open(path) as f:
            ALL_RUNS.update(json.load(f))


def new_run() -> dict:
    now = datetime.now()
    run = now.strftime("%Y_%m_%d--%H_%M_%S")
    m = {"RunName": run, "RunDate": now.strftime("%b %d, %Y %I:%M %p"), "latent_moe_metrics": {}}
    ALL_RUNS[run] = m
    return m


def save(path="./metrics/metrics.json", backup_dir="./metrics/backups/", jsons_path="./metrics/jsons/",
         run_metrics: Optional[dict] = None) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    os.makedirs(backup_dir, exist_ok=True)
    with open(path, "w") as f:
        json.dump(ALL_RUNS, f, indent=4)
    if run_metrics:
        os.makedirs(jsons_path, exist_ok=True)
        note = (run_metrics.get("note") or "")[:15]
        with open(os.path.join(jsons_path, f"{run_metrics['RunName']}--{note}.json"), "w") as f:
            json.dump(run_metrics, f, indent=4)
