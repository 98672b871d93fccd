"""Run the reference's own inference scripts UNCHANGED on this implementation.

    python -m longspec_amd.harness /path/to/LongSpec/longspec/test/inference_long-bench.py --model_name llama8b ...
    python -m longspec_amd.harness /path/to/LongSpec/longspec/test/inference_qwq.py ...

The scripts import their models as sibling modules (``from llama_glide import LlamaGlide``,
``inference_long-bench.py:1``; ``from qwen2_glide import Qwen2Glide``, ``inference_qwq.py:1``) and Python puts the
script's own directory first on ``sys.path``, so a PYTHONPATH entry cannot shadow them.  This launcher registers the
MI355X modules under those names in ``sys.modules`` *before* the script runs (an already-imported module always
wins) and then executes the script as ``__main__`` with its own argv.  Nothing in the reference tree is edited.
"""
from __future__ import annotations

import importlib
import runpy
import sys

# reference module name -> module of this package with the same public classes
ALIASES = {
    "llama_glide": "longspec_amd.llama_glide",
    "qwen2_glide": "longspec_amd.qwen2_glide",
    "llama": "longspec_amd.llama",
    "qwen2": "longspec_amd.qwen2",
}


def install_aliases() -> None:
    for ref_name, ours in ALIASES.items():
        sys.modules[ref_name] = importlib.import_module(ours)


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m longspec_amd.harness <reference script.py> [its arguments ...]")
    install_aliases()
    script = argv[0]
    sys.argv = argv                      # the script parses its own arguments
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
