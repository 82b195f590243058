#!/usr/bin/env python3
"""Fixture G5 (runs ONLY in the build container): the reference's own synthetic generator (benchmark/
generate_synthetic_data.py:345-470) writes its memmap cache for (S=1000, D=20, sparsity=0.2, seed=42, float32) into
tests/golden/g5_cache/ -- five raw .dat arrays + _meta.json.  The files are data; benchmark/cache_format.py must read them
and write byte-identical ones.  Re-run with:  python tests/golden/make_golden_cache.py"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
stub = tempfile.mkdtemp(prefix="mlflow_stub_")
os.makedirs(os.path.join(stub, "mlflow"), exist_ok=True)
open(os.path.join(stub, "mlflow", "__init__.py"), "w").close()
sys.path[:0] = [stub, os.path.join(REF, "src"), os.path.join(REF, "benchmark")]

import torch  # noqa: E402
from generate_synthetic_data import generate_synthetic_matching_input_args  # noqa: E402

out = os.path.join(HERE, "g5_cache")
shutil.rmtree(out, ignore_errors=True)
args = generate_synthetic_matching_input_args(1000, 20, 0.2, device="cpu", dtype=torch.float32, seed=42, cache_dir=out)
print(sorted(os.listdir(out)), "nnz", args.A.values().numel())
