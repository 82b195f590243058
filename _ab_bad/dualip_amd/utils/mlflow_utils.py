"""Run tracking: hyper-parameters once, a handful of scalars per iteration.

Public names and behaviour follow the reference's src/dualip/utils/mlflow_utils.py (MLflowConfig :11-21,
mlflow_run_context :55-91, log_hyperparameters :94-113, log_metrics :152-173, log_objective_result :176-203,
is_mlflow_available :206-213): tracking is off unless a config with ``enabled=True`` is active, and a tracking failure
is printed and never stops the solve.

Two things differ, both because of where the solve runs here:
  * the device-resident AGD loop keeps its per-iteration scalars in a device log and the host reads them in chunks, so
    metrics arrive through ``log_iteration_rows`` -- a block of iterations at a time, no per-iteration synchronisation;
  * when the ``mlflow`` package is not importable (it is not part of the ROCm image) the same calls write a plain file
    store instead of silently doing nothing:  <tracking_uri or ./dualip_runs>/<experiment>/<run>/params.json and
    metrics.csv (columns step,key,value -- what ``mlflow.log_metric`` would have received).
"""
import csv
import json
import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Dict, Optional, Sequence, Union

import torch

from dualip_amd.types import (
    LOG_DUAL_OBJECTIVE,
    LOG_MAX_POS_SLACK,
    LOG_PRIMAL_OBJECTIVE,
    LOG_REG_PENALTY,
    LOG_STEP_SIZE,
    LOG_SUM_POS_SLACK,
    ObjectiveResult,
)

SOLVER_PARAMS = ("max_iter", "initial_step_size", "max_step_size", "gamma", "gamma_decay_type")  # mlflow_utils.py:120
OBJECTIVE_PARAMS = ("objective_type",)  # :131


@dataclass
class MLflowConfig:
    enabled: bool
    tracking_uri: str = ""
    experiment_name: str = ""
    run_name: str = ""
    log_hyperparameters: bool = True
    log_metrics: bool = True
    synchronous: bool = False


def is_mlflow_available() -> bool:
    try:
        import mlflow  # noqa: F401

        return True
    except ImportError:
        return False


class _MlflowBackend:
    def __init__(self, config: MLflowConfig):
        import mlflow

        self.mlflow, self.config = mlflow, config
        if config.tracking_uri:
            mlflow.set_tracking_uri(config.tracking_uri)
        experiment = mlflow.set_experiment(config.experiment_name or "dualip_experiments")
        self._ctx = mlflow.start_run(run_name=config.run_name or "dualip_run", experiment_id=experiment.experiment_id)
        self.run = self._ctx.__enter__()
        print(f"Started MLflow run: {config.run_name or 'dualip_run'} id: {self.run.info.run_id}")

    def param(self, key, value):
        self.mlflow.log_param(key, value)

    def metrics(self, step, items):
        for key, value in items:
            self.mlflow.log_metric(key, value, step=step, synchronous=self.config.synchronous)

    def close(self):
        self._ctx.__exit__(None, None, None)


class _FileBackend:
    """params.json + metrics.csv in one directory per run (a second run of the same name gets a numeric suffix)."""

    def __init__(self, config: MLflowConfig):
        root = config.tracking_uri or "dualip_runs"
        if root.startswith("file:"):
            root = root[5:]
            root = "/" + root.lstrip("/") if root.startswith("//") else root
        base = os.path.join(root, config.experiment_name or "dualip_experiments", config.run_name or "dualip_run")
        path, k = base, 1
        while os.path.exists(path):
            path, k = f"{base}_{k}", k + 1
        os.makedirs(path)
        self.path = path
        self.run = path
        self._params: Dict[str, Any] = {}
        self._file = open(os.path.join(path, "metrics.csv"), "w", newline="")
        self._csv = csv.writer(self._file)
        self._csv.writerow(["step", "key", "value"])
        print(f"Tracking run in {path} (mlflow is not installed: file store)")

    def param(self, key, value):
        self._params[key] = value
        with open(os.path.join(self.path, "params.json"), "w") as fh:
            json.dump(self._params, fh, indent=2, sort_keys=True)

    def metrics(self, step, items):
        for key, value in items:
            self._csv.writerow(["" if step is None else int(step), key, repr(float(value))])

    def close(self):
        self._file.close()


class _State:
    def __init__(self):
        self.config: Optional[MLflowConfig] = None
        self.backend = None

    def is_enabled(self) -> bool:
        return self.backend is not None and self.config is not None and self.config.enabled


_state = _State()


def tracking_enabled() -> bool:
    """True inside an ``mlflow_run_context`` whose config is enabled (the AGD loop asks before fetching device logs)."""
    return _state.is_enabled() and bool(_state.config.log_metrics)


@contextmanager
def mlflow_run_context(config: MLflowConfig):
    """Open a run for the duration of the block; yields the run (mlflow Run object or the file store's directory) or None."""
    if config is None or not config.enabled:
        yield None
        return
    backend = None
    try:
        backend = _MlflowBackend(config) if is_mlflow_available() else _FileBackend(config)
    except Exception as e:  # tracking must never stop the solve (mlflow_utils.py:85-87)
        print(f"MLflow logging failed: {e}. Continuing without MLflow logging.")
    _state.config, _state.backend = config, backend
    try:
        yield backend.run if backend is not None else None
    finally:
        _state.config, _state.backend = None, None
        if backend is not None:
            try:
                backend.close()
            except Exception as e:
                print(f"MLflow logging failed: {e}.")


def _plain(value: Any):
    if isinstance(value, (int, float, str, bool)):
        return value
    if isinstance(value, torch.Tensor):
        return value.item() if value.numel() == 1 else None
    return str(value)


def log_hyperparameters(params: Dict[str, Any], step: Optional[int] = None) -> None:
    """``params = {"solver": {...}, "objective": {...}}``; the reference's selection of keys is logged as
    ``solver.<key>`` / ``objective.<key>``."""
    if not _state.is_enabled() or not _state.config.log_hyperparameters:
        return
    try:
        for group, keep in (("solver", SOLVER_PARAMS), ("objective", OBJECTIVE_PARAMS)):
            for key, value in (params.get(group) or {}).items():
                if key in keep:
                    value = _plain(value)
                    if value is not None:
                        _state.backend.param(f"{group}.{key}", value)
    except Exception as e:
        print(f"Failed to log hyperparameters: {e}")


def log_metrics(metrics: Dict[str, Union[float, int]], step: Optional[int] = None) -> None:
    if not tracking_enabled():
        return
    try:
        items = []
        for key, value in metrics.items():
            if isinstance(value, (int, float, bool)):
                items.append((key, value))
            else:
                print(f"Skipped metric {key} (type: {type(value).__name__})")
        _state.backend.metrics(step, items)
    except Exception as e:
        print(f"Failed to log metrics: {e}")


_RESULT_FIELDS = (  # attribute of ObjectiveResult -> metric name (mlflow_utils.py:186-198)
    ("dual_objective", "dual_objective"),
    ("primal_objective", "primal_objective"),
    ("reg_penalty", "regularization_penalty"),
    ("max_pos_slack", "max_positive_slack"),
    ("sum_pos_slack", "sum_positive_slack"),
)


def log_objective_result(result: ObjectiveResult, step: Optional[int] = None) -> None:
    if not tracking_enabled():
        return
    try:
        metrics = {}
        for attr, name in _RESULT_FIELDS:
            value = getattr(result, attr, None)
            if value is not None:
                metrics[name] = value.item() if hasattr(value, "item") else float(value)
        if metrics:
            log_metrics(metrics, step)
    except Exception as e:
        print(f"Failed to log objective result: {e}")


def log_iteration_rows(first_iteration: int, rows: Sequence[Sequence[float]], gammas: Optional[Sequence[float]] = None, with_primal_last: bool = False) -> None:
    """Block form for the device-resident loop: ``rows[k]`` is the device log row (types.LOG_*) of iteration
    ``first_iteration + k``; emits per iteration the metrics the reference logs at agd.py:190-201 (step_size,
    dual_objective, gamma, then the objective's scalars)."""
    if not tracking_enabled():
        return
    try:
        for k, row in enumerate(rows):
            step = first_iteration + k
            items = [("step_size", float(row[LOG_STEP_SIZE])), ("dual_objective", float(row[LOG_DUAL_OBJECTIVE]))]
            if gammas is not None:
                items.append(("gamma", float(gammas[k])))
            if with_primal_last and k == len(rows) - 1:
                items.append(("primal_objective", float(row[LOG_PRIMAL_OBJECTIVE])))
            items += [
                ("regularization_penalty", float(row[LOG_REG_PENALTY])),
                ("max_positive_slack", float(row[LOG_MAX_POS_SLACK])),
                ("sum_positive_slack", float(row[LOG_SUM_POS_SLACK])),
            ]
            _state.backend.metrics(step, items)
    except Exception as e:
        print(f"Failed to log metrics: {e}")
