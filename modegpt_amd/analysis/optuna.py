"""Hyper-parameter sweep over the ridges and the sparsity smoothing (reference: src/analysis/optuna.py:16-48):
every trial runs the whole driver with a sampled CompressionConfig and reports the compressed perplexity to minimise.

    python -m modegpt_amd.analysis.optuna --model <hf model> --compression_ratio 0.3 --n_trials 20

`optuna` is imported lazily: the package is optional (it is not installed in the build image).  Each trial costs one
full calibration + compression; on one MI355X that is minutes for an 8B model, which is what makes a sweep practical.
"""
from __future__ import annotations

import argparse
import dataclasses


def objective_factory(base_args):
    from ..adapters.CompressionConfig import CompressionConfig
    from ..run_modegpt import main

    def objective(trial):
        cfg = CompressionConfig.from_args(base_args)
        cfg = dataclasses.replace(
            cfg,
            nystrom_ridge=trial.suggest_float("nystrom_ridge", 1e-6, 1e-1, log=True),
            ridge_vo=trial.suggest_float("ridge_vo", 1e-7, 1e-2, log=True),
            ridge_qk=trial.suggest_float("ridge_qk", 1e-7, 1e-1, log=True),
            sparsity_smoothing=trial.suggest_float("sparsity_smoothing", 0.01, 0.5, log=True),
            note=f"optuna-{trial.number}",
        )
        return main(trial, config=cfg)

    return objective


def run(argv=None):
    try:
        import optuna
    except ImportError as e:  # fail loudly with the remedy, never silently skip
        raise SystemExit("modegpt_amd.analysis.optuna needs the `optuna` package (pip install optuna)") from e
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--n_trials", type=int, default=20)
    ap.add_argument("--storage", default="sqlite:///modegpt_optuna.db")
    ap.add_argument("--study", default="modegpt")
    own, rest = ap.parse_known_args(argv)
    study = optuna.create_study(study_name=own.study, storage=own.storage, direction="minimize", load_if_exists=True)
    study.optimize(objective_factory(rest), n_trials=own.n_trials)
    print("best", study.best_value, study.best_params)
    return study


if __name__ == "__main__":
    run()
