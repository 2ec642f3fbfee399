"""Golden vectors for the pose-error / cm-degree bookkeeping, produced by RUNNING THE REFERENCE
(src/evaluators/cmd_evaluator.py::Evaluator; src/utils/eval_utils.py::query_pose_error is not importable here because its
module imports cv2, so its two numbers are taken from the identical expressions inside the Evaluator's metrics).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_eval_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("ONEPOSE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from src.evaluators.cmd_evaluator import Evaluator  # noqa: E402  (reference)
from onepose_amd import synthetic  # noqa: E402


def main():
    ev = Evaluator()
    seq = synthetic.make_pose_pairs(0)
    for i, (pred, gt) in enumerate(seq):
        ev.evaluate(None if i == 5 else pred, gt)
    hits = {"cmd1": [bool(x) for x in ev.cmd1], "cmd3": [bool(x) for x in ev.cmd3], "cmd5": [bool(x) for x in ev.cmd5]}
    summary = {k: float(v) for k, v in ev.summarize().items()}
    with open(os.path.join(HERE, "eval_golden.json"), "w") as f:
        json.dump({"seed": 0, "none_at": 5, "hits": hits, "summary": summary}, f, indent=1)
    print(summary)


if __name__ == "__main__":
    main()
